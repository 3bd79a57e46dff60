// mrt_cull2.hip -- two-level tile culling for scenes with many primitives (C3: 10k spheres, C4: 1M triangles).
//
// mrt_cull.hip makes every 8x8 tile read the bound of EVERY primitive.  Here a workgroup owns a super-tile
// (4 x 4 tiles with 4 waves, 8 x 4 tiles with 8 waves; every wave owns 4 horizontally adjacent tiles):
//
//   level 1  all threads of the workgroup walk a chunk of the primitive list together, one primitive per
//            thread (coalesced SoA reads), test it against the bound of the WHOLE super-tile's rays and append
//            the survivors' indices to a list in LDS (ballot + popcount compaction, one LDS atomic per wave);
//   level 2  every wave, for each of its 4 tiles, tests the short LDS list against the tile's own bound
//            (one survivor per lane), ballots, and runs the exact reference test on what is left.
//
// The same two levels serve the shadow rays of each light (super bundle = ball around every hit point of
// the workgroup, tile bundle = ball around the tile's hit points, swept towards the light).
// Results are those of the brute-force loops (mrt_kernels.hip) bit for bit: culling is conservative
// (mrt_cull_common.h) and the closest-hit update does not depend on the visiting order.

#include "mrt_internal.h"
#include "mrt_device.h"
#include "mrt_cull_common.h"

#include <cmath>
#include <cstdlib>

namespace mrt {

constexpr int CULL2_CAP = 8192;   // capacity of the LDS survivor list (32 KiB)
constexpr int CULL2_SPAN = 16384; // primitives per level-1 pass; a pass whose survivors overflow the list is
                                  // redone in spans of CULL2_CAP, which always fit
constexpr int TPW = 4;          // tiles per wave

// append the lanes with pass == true to the LDS list
__device__ __forceinline__ void list_append(int *list, int *count, bool pass, int value, int lane)
{
  const unsigned long long mask = __ballot(pass);
  if (mask == 0) return;
  int base = 0;
  if (lane == 0) base = atomicAdd(count, __popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);
  const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
  if (pass && slot < CULL2_CAP) list[slot] = value; // the counter still counts: overflow is detected by the caller
}

// Level 1 over primitives [c0, cend): every thread of the workgroup takes one primitive per step, 4 steps
// per iteration with all 16 loads issued before the first test (the loop is latency-bound otherwise).
// bx..br are the bound streams (sphere centre + radius, or triangle bounding sphere); test(x,y,z,r) -> pass.
template <int NT, class Test>
__device__ __forceinline__ void level1_pass(const float *__restrict__ bx, const float *__restrict__ by,
                                            const float *__restrict__ bz, const float *__restrict__ br, int c0,
                                            int cend, int wave, int lane, int *list, int *count, Test test)
{
  constexpr int U = 4;
  for (int b = c0 + wave * 64; b < cend; b += NT * U) {
    float x[U], y[U], z[U], r[U];
    int idx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      idx[u] = b + u * NT + lane;
      const int ii = min(idx[u], cend - 1);
      x[u] = bx[ii]; y[u] = by[ii]; z[u] = bz[ii]; r[u] = br[ii];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool pass = (idx[u] < cend) && test(x[u], y[u], z[u], r[u]);
      list_append(list, count, pass, idx[u], lane);
    }
  }
}

struct Cone { float ax, ay, az, cosT, sinT; };

// cone around the rays of the screen rectangle centred at (um, vm) (in u,v) whose rays differ from the
// centre ray by at most ext
__device__ __forceinline__ Cone make_cone(const DevCamera &cam, float um, float vm, float ext)
{
  Cone c;
  c.ax = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
  c.ay = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
  c.az = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
  const float inv = fast_rsq(dot3(c.ax, c.ay, c.az, c.ax, c.ay, c.az));
  c.ax *= inv; c.ay *= inv; c.az *= inv;
  c.sinT = (ext + 2.0f * camera_rounding(cam)) * inv * (1.0f + 1e-4f) + 1e-6f; // + the rounding of ray and centre
  c.cosT = fast_sqrt(fmaxf(1.0f - c.sinT * c.sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
  if (!(c.sinT < 0.99f) || !(c.cosT > 0.0f)) { c.cosT = -1.0f; c.sinT = 0.0f; } // cannot cull
  return c;
}

// max |+-du*hor +- dv*ver|
__device__ __forceinline__ float rect_ext(const DevCamera &cam, float du, float dv)
{
  const float ax = du * cam.hx, ay = du * cam.hy, az = du * cam.hz;
  const float bx = dv * cam.vx, by = dv * cam.vy, bz = dv * cam.vz;
  const float px = ax + bx, py = ay + by, pz = az + bz, mx = ax - bx, my = ay - by, mz = az - bz;
  return fast_sqrt(fmaxf(dot3(px, py, pz, px, py, pz), dot3(mx, my, mz, mx, my, mz))) * 1.001f;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void render_cull2_kernel(RenderArgs A)
{
  constexpr int NT = WAVES * 64;
  constexpr int STRIPS = WAVES / 4;  // wave strips (4 tiles each) per super-tile row: super-tile = 4*STRIPS x 4 tiles
  __shared__ int s_list[CULL2_CAP];
  __shared__ int s_count;
  __shared__ float s_red[WAVES][8];

  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;

  const int wrow = wave / STRIPS, wcol = wave - wrow * STRIPS;
  const int tile_y = blockIdx.y * 4 + wrow;
  const int tile_x0 = (blockIdx.x * STRIPS + wcol) * TPW;
  const int lrow = tile_y * 8 + ty;
  // global rows: local rows map to strips dealt round-robin over ranks (a tile never straddles two strips)
  auto global_row0 = [&](int trow) {
    const int l0 = trow * 8;
    const int strip = l0 / A.strip_rows;
    return (strip * A.nranks + A.rank) * A.strip_rows + (l0 - strip * A.strip_rows);
  };
  const int grow0 = global_row0(tile_y);
  const int grow = grow0 + ty;
  const bool row_ok = (lrow < A.local_rows) && (grow < A.H);

  const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H));
  // super-tile cone: columns [x0*8-1, x0*8 + 32*STRIPS], global rows of the first to the last tile row
  Cone SC;
  {
    const int last_trow = min(blockIdx.y * 4 + 3, max((A.local_rows + 7) / 8 - 1, 0));
    const float c_lo = float(blockIdx.x * STRIPS * TPW * 8 - 1), c_hi = float((blockIdx.x + 1) * STRIPS * TPW * 8);
    const float r_lo = float(global_row0(blockIdx.y * 4) - 1), r_hi = float(global_row0(last_trow) + 8);
    const float du = 0.5f * (c_hi - c_lo) * rw, dv = 0.5f * (r_hi - r_lo) * rh;
    SC = make_cone(cam, 0.5f * (c_lo + c_hi) * rw, 0.5f * (r_lo + r_hi) * rh, rect_ext(cam, du, dv));
  }
  const float tile_ext = rect_ext(cam, 4.5f * rw, 4.5f * rh);

  bool valid[TPW];
  int col[TPW];
  Cone TC[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    col[k] = (tile_x0 + k) * 8 + tx;
    valid[k] = row_ok && (col[k] < A.W);
    TC[k] = make_cone(cam, (float((tile_x0 + k) * 8) + 3.5f) * rw, (float(grow0) + 3.5f) * rh, tile_ext);
  }
  const float v = float(grow) / float(A.H); // src/main.cu:137

  float accr[TPW], accg[TPW], accb[TPW], out_t[TPW];
  int out_id[TPW];
#pragma unroll
  for (int k = 0; k < TPW; ++k) { accr[k] = accg[k] = accb[k] = 0.0f; out_id[k] = -1; out_t[k] = MRT_INF; }
  unsigned int n_pix = 0, n_shadow = 0, n_hit = 0, n_occl = 0;
#pragma unroll
  for (int k = 0; k < TPW; ++k) n_pix += valid[k] ? 1u : 0u;

  const int nsamp = A.spp;
  const float DIVW = 13.0f, DIVH = 13.0f;

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) { // pipeline/multisample_mesh.cuh:23-30
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
    // ---- primary rays: pipeline/viewport.cuh:30-34 ----
    float dx[TPW], dy[TPW], dz[TPW], aa[TPW], best_t[TPW];
    int best_obj[TPW], best_prim[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      float su = float(col[k]) / float(A.W), sv = v; // src/main.cu:136
      if (nsamp > 1) { su = su + offx; sv = v + offy; }
      dx[k] = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
      dy[k] = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      dz[k] = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
      aa[k] = dot3(dx[k], dy[k], dz[k], dx[k], dy[k], dz[k]);
      best_t[k] = MRT_INF; best_obj[k] = 0x7fffffff; best_prim[k] = -1;
    }

    // ---- closest hit over spheres: src/main.cu:60-77, core/sphere.cuh:32-73, interval [0.01, inf] ----
    for (int c0 = 0, span = CULL2_SPAN; c0 < ns;) {
      const int cend = min(ns, c0 + span);
      if (tid == 0) s_count = 0;
      __syncthreads();
      level1_pass<NT>(sc.sph_cx, sc.sph_cy, sc.sph_cz, sc.sph_r, c0, cend, wave, lane, s_list, &s_count,
                      [&](float cx, float cy, float cz, float r) {
                        const float wx = cx - cam.ox, wy = cy - cam.oy, wz = cz - cam.oz;
                        return ball_may_touch_cone(wx, wy, wz, sphere_cull_radius(r * r, dot3(wx, wy, wz, wx, wy, wz)),
                                                   SC.ax, SC.ay, SC.az, SC.cosT, SC.sinT);
                      });
      __syncthreads();
      const int cnt = s_count;
      if (cnt > CULL2_CAP) { span = CULL2_CAP; __syncthreads(); continue; } // overflow: redo in spans that fit
      for (int lb = 0; lb < cnt; lb += 64) { // level 2
        const int e = lb + lane;
        int idx = 0;
        float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f, reff = 0.0f, wx = 0.0f, wy = 0.0f, wz = 0.0f;
        if (e < cnt) {
          idx = s_list[e];
          cx = sc.sph_cx[idx]; cy = sc.sph_cy[idx]; cz = sc.sph_cz[idx]; r = sc.sph_r[idx];
          wx = cx - cam.ox; wy = cy - cam.oy; wz = cz - cam.oz;
          reff = sphere_cull_radius(r * r, dot3(wx, wy, wz, wx, wy, wz));
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const bool pass = (e < cnt) && ball_may_touch_cone(wx, wy, wz, reff, TC[k].ax, TC[k].ay, TC[k].az, TC[k].cosT,
                                                             TC[k].sinT);
          unsigned long long mask = __ballot(pass);
          while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float scx = bcast(cx, j), scy = bcast(cy, j), scz = bcast(cz, j), sr = bcast(r, j);
            const float ocx = cam.ox - scx, ocy = cam.oy - scy, ocz = cam.oz - scz;
            const float b = dot3(ocx, ocy, ocz, dx[k], dy[k], dz[k]);
            const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
            const float disc = b * b - aa[k] * c;
            if (disc >= 0.0f) {
              float t;
              if (sphere_roots(aa[k], b, disc, 0.01f, MRT_INF, t)) {
                const int g = __builtin_amdgcn_readlane(idx, j);
                const int obj = sc.sph_obj[g];
                if (t < best_t[k] || (t == best_t[k] && t < MRT_INF && obj < best_obj[k])) {
                  best_t[k] = t; best_obj[k] = obj; best_prim[k] = g;
                }
              }
            }
          }
        }
      }
      __syncthreads();
      c0 = cend; span = CULL2_SPAN;
    }

    // ---- closest hit over triangles: core/triangle.cuh:77-103 ----
    for (int c0 = 0, span = CULL2_SPAN; c0 < nt;) {
      const int cend = min(nt, c0 + span);
      if (tid == 0) s_count = 0;
      __syncthreads();
      level1_pass<NT>(sc.tri_bx, sc.tri_by, sc.tri_bz, sc.tri_br, c0, cend, wave, lane, s_list, &s_count,
                      [&](float bx, float by, float bz, float br) {
                        return ball_may_touch_cone(bx - cam.ox, by - cam.oy, bz - cam.oz, br * CULL_REL + CULL_ABS, SC.ax,
                                                   SC.ay, SC.az, SC.cosT, SC.sinT);
                      });
      __syncthreads();
      const int cnt = s_count;
      if (cnt > CULL2_CAP) { span = CULL2_CAP; __syncthreads(); continue; }
      for (int lb = 0; lb < cnt; lb += 64) { // level 2
        const int e = lb + lane;
        int idx = 0;
        float wx = 0.0f, wy = 0.0f, wz = 0.0f, reff = 0.0f;
        if (e < cnt) {
          idx = s_list[e];
          wx = sc.tri_bx[idx] - cam.ox; wy = sc.tri_by[idx] - cam.oy; wz = sc.tri_bz[idx] - cam.oz;
          reff = sc.tri_br[idx] * CULL_REL + CULL_ABS;
        }
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const bool pass = (e < cnt) && ball_may_touch_cone(wx, wy, wz, reff, TC[k].ax, TC[k].ay, TC[k].az, TC[k].cosT,
                                                             TC[k].sinT);
          unsigned long long mask = __ballot(pass);
          while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int g = __builtin_amdgcn_readlane(idx, j);
            const float *q = sc.tri_rec + 16 * size_t(g); // wave-uniform: one s_load_dwordx16
            const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
            const float rdn = dot3(dx[k], dy[k], dz[k], pnx, pny, pnz);
            const float t = (kp - dot3(cam.ox, cam.oy, cam.oz, pnx, pny, pnz)) / rdn;
            const float px = cam.ox + t * dx[k], py = cam.oy + t * dy[k], pz = cam.oz + t * dz[k];
            const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
            const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
            const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
            const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
            if (hit) {
              const int obj = sc.tri_obj[g];
              if (t < best_t[k] || (t == best_t[k] && t < MRT_INF && obj < best_obj[k])) {
                best_t[k] = t; best_obj[k] = obj; best_prim[k] = ns + g;
              }
            }
          }
        }
      }
      __syncthreads();
      c0 = cend; span = CULL2_SPAN;
    }

    // ---- hit records (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
    bool have_hit[TPW];
    float hpx[TPW], hpy[TPW], hpz[TPW], hnx[TPW], hny[TPW], hnz[TPW];
    bool any_hit = false;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      have_hit[k] = best_prim[k] >= 0;
      hpx[k] = hpy[k] = hpz[k] = hnx[k] = hny[k] = hnz[k] = 0.0f;
      if (have_hit[k]) {
        hpx[k] = cam.ox + best_t[k] * dx[k]; hpy[k] = cam.oy + best_t[k] * dy[k]; hpz[k] = cam.oz + best_t[k] * dz[k];
        if (best_prim[k] < ns) {
          const int g = best_prim[k];
          float nx = hpx[k] - sc.sph_cx[g], ny = hpy[k] - sc.sph_cy[g], nz = hpz[k] - sc.sph_cz[g];
          normalize3(nx, ny, nz);
          hnx[k] = nx; hny[k] = ny; hnz[k] = nz;
        } else {
          const float *q = sc.tri_rec + 16 * size_t(best_prim[k] - ns);
          const float pnx = q[0], pny = q[1], pnz = q[2];
          const bool front = dot3(dx[k], dy[k], dz[k], pnx, pny, pnz) < 0;
          hnx[k] = front ? pnx : -pnx; hny[k] = front ? pny : -pny; hnz[k] = front ? pnz : -pnz;
        }
        if (valid[k]) n_hit++;
        any_hit = true;
      }
    }

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr[TPW], lg[TPW], lb[TPW];
    bool have_light[TPW];
#pragma unroll
    for (int k = 0; k < TPW; ++k) { lr[k] = lg[k] = lb[k] = 0.0f; have_light[k] = false; }

    if (__syncthreads_or(any_hit ? 1 : 0)) { // workgroup-uniform: sky-only super-tiles skip the lights
      // tile bundles (per wave) and the super bundle (workgroup) around the hit points
      const float big = 3.0e38f;
      float tmx[TPW], tmy[TPW], tmz[TPW], trho[TPW];
      float wlo[3] = {big, big, big}, whi[3] = {-big, -big, -big};
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const float lox = wave_min(have_hit[k] ? hpx[k] : big), hix = wave_max(have_hit[k] ? hpx[k] : -big);
        const float loy = wave_min(have_hit[k] ? hpy[k] : big), hiy = wave_max(have_hit[k] ? hpy[k] : -big);
        const float loz = wave_min(have_hit[k] ? hpz[k] : big), hiz = wave_max(have_hit[k] ? hpz[k] : -big);
        tmx[k] = 0.5f * (lox + hix); tmy[k] = 0.5f * (loy + hiy); tmz[k] = 0.5f * (loz + hiz);
        const float ex = hpx[k] - tmx[k], ey = hpy[k] - tmy[k], ez = hpz[k] - tmz[k];
        trho[k] = fast_sqrt(wave_max(have_hit[k] ? dot3(ex, ey, ez, ex, ey, ez) : 0.0f)) * CULL_REL + CULL_ABS;
        wlo[0] = fminf(wlo[0], lox); wlo[1] = fminf(wlo[1], loy); wlo[2] = fminf(wlo[2], loz);
        whi[0] = fmaxf(whi[0], hix); whi[1] = fmaxf(whi[1], hiy); whi[2] = fmaxf(whi[2], hiz);
      }
      if (lane == 0) {
        s_red[wave][0] = wlo[0]; s_red[wave][1] = wlo[1]; s_red[wave][2] = wlo[2];
        s_red[wave][3] = whi[0]; s_red[wave][4] = whi[1]; s_red[wave][5] = whi[2];
      }
      __syncthreads();
      float slo[3] = {big, big, big}, shi[3] = {-big, -big, -big};
#pragma unroll
      for (int w = 0; w < WAVES; ++w)
#pragma unroll
        for (int a = 0; a < 3; ++a) { slo[a] = fminf(slo[a], s_red[w][a]); shi[a] = fmaxf(shi[a], s_red[w][3 + a]); }
      Bundle SB;
      SB.mx = 0.5f * (slo[0] + shi[0]); SB.my = 0.5f * (slo[1] + shi[1]); SB.mz = 0.5f * (slo[2] + shi[2]);
      {
        float e2 = 0.0f;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          const float ex = hpx[k] - SB.mx, ey = hpy[k] - SB.my, ez = hpz[k] - SB.mz;
          e2 = fmaxf(e2, have_hit[k] ? dot3(ex, ey, ez, ex, ey, ez) : 0.0f);
        }
        e2 = wave_max(e2);
        if (lane == 0) s_red[wave][6] = e2;
        __syncthreads();
        float m2 = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) m2 = fmaxf(m2, s_red[w][6]);
        SB.rho = fast_sqrt(m2) * CULL_REL + CULL_ABS;
      }

      for (int li = 0; li < sc.nl; ++li) {
        const int lkind = sc.light_kind[li];
        const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
        float sdx[TPW], sdy[TPW], sdz[TPW], sa[TPW], tl[TPW], lnx[TPW], lny[TPW], lnz[TPW];
        bool need[TPW], open[TPW];
        Bundle TB[TPW];
        bool any_open = false;
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          TB[k].mx = tmx[k]; TB[k].my = tmy[k]; TB[k].mz = tmz[k]; TB[k].rho = trho[k];
          if (lkind == 0) { // DirectionalLight::ray_from, core/light.cuh:73-81
            sdx[k] = -lvx; sdy[k] = -lvy; sdz[k] = -lvz; tl[k] = MRT_INF;
            lnx[k] = lvx; lny[k] = lvy; lnz[k] = lvz;
            TB[k].point = false; TB[k].Dx = -lvx; TB[k].Dy = -lvy; TB[k].Dz = -lvz;
          } else {          // PointLight::ray_from, core/light.cuh:47-53
            sdx[k] = lvx - hpx[k]; sdy[k] = lvy - hpy[k]; sdz[k] = lvz - hpz[k]; tl[k] = 1.0f;
            lnx[k] = hpx[k] - lvx; lny[k] = hpy[k] - lvy; lnz[k] = hpz[k] - lvz;
            normalize3(lnx[k], lny[k], lnz[k]);
            TB[k].point = true; TB[k].Dx = lvx - tmx[k]; TB[k].Dy = lvy - tmy[k]; TB[k].Dz = lvz - tmz[k];
          }
          TB[k].DD = dot3(TB[k].Dx, TB[k].Dy, TB[k].Dz, TB[k].Dx, TB[k].Dy, TB[k].Dz);
          sa[k] = dot3(sdx[k], sdy[k], sdz[k], sdx[k], sdy[k], sdz[k]);
          // the light must not shine through the surface, core/light.cuh:148
          need[k] = have_hit[k] && !(dot3(sdx[k], sdy[k], sdz[k], hnx[k], hny[k], hnz[k]) < 0);
          open[k] = need[k];
          any_open |= open[k];
        }
        if (lkind == 0) { SB.point = false; SB.Dx = -lvx; SB.Dy = -lvy; SB.Dz = -lvz; }
        else { SB.point = true; SB.Dx = lvx - SB.mx; SB.Dy = lvy - SB.my; SB.Dz = lvz - SB.mz; }
        SB.DD = dot3(SB.Dx, SB.Dy, SB.Dz, SB.Dx, SB.Dy, SB.Dz);

        // any-hit over spheres, interval [1e-3, t_light] (core/light.cuh:28-33)
        for (int c0 = 0, span = CULL2_SPAN; c0 < ns;) {
          any_open = false;
#pragma unroll
          for (int k = 0; k < TPW; ++k) any_open |= open[k];
          if (tid == 0) s_count = 0;
          if (!__syncthreads_or(any_open ? 1 : 0)) break; // nobody in the workgroup has an undecided ray
          const int cend = min(ns, c0 + span);
          level1_pass<NT>(sc.sph_cx, sc.sph_cy, sc.sph_cz, sc.sph_r, c0, cend, wave, lane, s_list, &s_count,
                          [&](float cx, float cy, float cz, float r) {
                            const float wx = cx - SB.mx, wy = cy - SB.my, wz = cz - SB.mz;
                            const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + SB.rho;
                            return bundle_may_touch(SB, cx, cy, cz, sphere_cull_radius(r * r, Lm * Lm));
                          });
          __syncthreads();
          const int cnt = s_count;
          if (cnt > CULL2_CAP) { span = CULL2_CAP; __syncthreads(); continue; }
          for (int lb = 0; lb < cnt; lb += 64) { // level 2
            const int e = lb + lane;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f;
            if (e < cnt) {
              const int idx = s_list[e];
              cx = sc.sph_cx[idx]; cy = sc.sph_cy[idx]; cz = sc.sph_cz[idx]; r = sc.sph_r[idx];
            }
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
              if (!__any(open[k])) continue;
              bool pass = false;
              if (e < cnt) {
                const float wx = cx - TB[k].mx, wy = cy - TB[k].my, wz = cz - TB[k].mz;
                const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + TB[k].rho;
                pass = bundle_may_touch(TB[k], cx, cy, cz, sphere_cull_radius(r * r, Lm * Lm));
              }
              unsigned long long mask = __ballot(pass);
              while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float scx = bcast(cx, j), scy = bcast(cy, j), scz = bcast(cz, j), sr = bcast(r, j);
                const float ocx = hpx[k] - scx, ocy = hpy[k] - scy, ocz = hpz[k] - scz;
                const float b = dot3(ocx, ocy, ocz, sdx[k], sdy[k], sdz[k]);
                const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
                const float disc = b * b - sa[k] * c;
                if (open[k] && disc >= 0.0f) {
                  float t;
                  if (sphere_roots(sa[k], b, disc, 1e-3f, tl[k], t)) open[k] = false;
                }
                if (!__any(open[k])) break; // ballot early-out for this tile
              }
            }
          }
          __syncthreads();
          c0 = cend; span = CULL2_SPAN;
        }
        // any-hit over triangles
        for (int c0 = 0, span = CULL2_SPAN; c0 < nt;) {
          any_open = false;
#pragma unroll
          for (int k = 0; k < TPW; ++k) any_open |= open[k];
          if (tid == 0) s_count = 0;
          if (!__syncthreads_or(any_open ? 1 : 0)) break;
          const int cend = min(nt, c0 + span);
          level1_pass<NT>(sc.tri_bx, sc.tri_by, sc.tri_bz, sc.tri_br, c0, cend, wave, lane, s_list, &s_count,
                          [&](float bx, float by, float bz, float br) { return bundle_may_touch(SB, bx, by, bz, br); });
          __syncthreads();
          const int cnt = s_count;
          if (cnt > CULL2_CAP) { span = CULL2_CAP; __syncthreads(); continue; }
          for (int lb = 0; lb < cnt; lb += 64) { // level 2
            const int e = lb + lane;
            int idx = 0;
            float bx = 0.0f, by = 0.0f, bz = 0.0f, br = 0.0f;
            if (e < cnt) {
              idx = s_list[e];
              bx = sc.tri_bx[idx]; by = sc.tri_by[idx]; bz = sc.tri_bz[idx]; br = sc.tri_br[idx];
            }
#pragma unroll
            for (int k = 0; k < TPW; ++k) {
              if (!__any(open[k])) continue;
              const bool pass = (e < cnt) && bundle_may_touch(TB[k], bx, by, bz, br);
              unsigned long long mask = __ballot(pass);
              while (mask) {
                const int j = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float *q = sc.tri_rec + 16 * size_t(__builtin_amdgcn_readlane(idx, j));
                const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
                const float rdn = dot3(sdx[k], sdy[k], sdz[k], pnx, pny, pnz);
                const float t = (kp - dot3(hpx[k], hpy[k], hpz[k], pnx, pny, pnz)) / rdn;
                const float px = hpx[k] + t * sdx[k], py = hpy[k] + t * sdy[k], pz = hpz[k] + t * sdz[k];
                const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
                const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
                const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
                const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl[k]);
                if (open[k] && h) open[k] = false;
                if (!__any(open[k])) break;
              }
            }
          }
          __syncthreads();
          c0 = cend; span = CULL2_SPAN;
        }

        // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168)
        const float lcr = sc.light_rgb[3 * li], lcg = sc.light_rgb[3 * li + 1], lcb = sc.light_rgb[3 * li + 2];
#pragma unroll
        for (int k = 0; k < TPW; ++k) {
          if (need[k] && valid[k]) { n_shadow++; if (!open[k]) n_occl++; }
          if (open[k]) {
            const float dn = dot3(sdx[k], sdy[k], sdz[k], lnx[k], lny[k], lnz[k]);
            float cr, cg, cb;
            if (sc.mat_kind[best_obj[k]] == 0) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
              const float scale = -dn;
              if (scale > 0.0f) { cr = scale * lcr; cg = scale * lcg; cb = scale * lcb; }
              else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
            } else {                             // MetallicMaterial::light_bounce, core/material.cuh:134-158
              const float s2 = 2.0f * dn;
              const float rx = sdx[k] - s2 * lnx[k], ry = sdy[k] - s2 * lny[k], rz = sdz[k] - s2 * lnz[k];
              if (dot3(rx, ry, rz, dx[k], dy[k], dz[k]) < -0.99f) { cr = lcr; cg = lcg; cb = lcb; }
              else { cr = 0.001f; cg = 0.001f; cb = 0.001f; }
            }
            if (have_light[k]) { lr[k] += cr; lg[k] += cg; lb[k] += cb; }
            else { lr[k] = cr; lg[k] = cg; lb[k] = cb; have_light[k] = true; }
          }
        }
      }
    }

    // ---- combine: src/main.cu:91-104 with MAX_RECURSION_DEPTH = 0 ----
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      float cr, cg, cb;
      if (have_hit[k]) {
        const float *m = sc.mat_rgb + 3 * best_obj[k];
        const float mr = m[0], mg = m[1], mb = m[2];
        const float inc = 0.01f; // what the depth-1 call returns, src/main.cu:51-53
        if (have_light[k]) { cr = inc * mr + lr[k] * mr; cg = inc * mg + lg[k] * mg; cb = inc * mb + lb[k] * mb; }
        else { cr = inc * mr; cg = inc * mg; cb = inc * mb; }
      } else {
        float nx = dx[k], ny = dy[k], nz = dz[k];
        normalize3(nx, ny, nz);
        const float y = 0.5f * (ny + 1.0f);
        const float w0 = 1.0f - y;
        cr = w0 * 1.0f + y * 0.5f; cg = w0 * 1.0f + y * 0.7f; cb = w0 * 1.0f + y * 1.0f;
      }
      accr[k] += cr; accg[k] += cg; accb[k] += cb; // src/main.cu:151
      out_id[k] = have_hit[k] ? best_obj[k] : -1;
      out_t[k] = best_t[k];
    }
  }

  const float fn = float(nsamp);
#pragma unroll
  for (int k = 0; k < TPW; ++k) {
    if (valid[k]) { // src/main.cu:155
      const size_t pix = size_t(lrow) * A.W + col[k];
      A.rgb[3 * pix + 0] = accr[k] / fn;
      A.rgb[3 * pix + 1] = accg[k] / fn;
      A.rgb[3 * pix + 2] = accb[k] / fn;
      if (A.id) A.id[pix] = out_id[k];
      if (A.t) A.t[pix] = out_t[k];
    }
  }

  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * ((blockIdx.x + 7u * blockIdx.y) & (unsigned)(A.stats_slots - 1));
    unsigned int v4[4] = {n_pix * (unsigned)nsamp, n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int x = v4[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[k], (unsigned long long)x);
    }
  }
}

hipError_t launch_render_cull2(const RenderArgs &A, int waves, hipStream_t stream)
{
  const int tiles_x = (A.W + 7) / 8, tiles_y = (A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  if (waves == 8) {
    dim3 grid((tiles_x + 7) / 8, (tiles_y + 3) / 4);
    hipLaunchKernelGGL(render_cull2_kernel<8>, grid, dim3(512), 0, stream, A);
  } else {
    dim3 grid((tiles_x + 3) / 4, (tiles_y + 3) / 4);
    hipLaunchKernelGGL(render_cull2_kernel<4>, grid, dim3(256), 0, stream, A);
  }
  return hipGetLastError();
}

} // namespace mrt
