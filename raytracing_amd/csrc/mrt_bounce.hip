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
// Bounce rays of neighbouring pixels diverge, so there is no tile to cull against: every ray walks every
// primitive (wave-uniform loop index: sphere streams and triangle records come through scalar loads), exactly
// the reference's loops.  One wave64 per 8x8 tile, same strip partition as the other render kernels.

#include "mrt_internal.h"
#include "mrt_device.h"

namespace mrt {

constexpr int MAX_LEVELS = 9; // depths 0..8

__global__ __launch_bounds__(256) void render_bounce_kernel(RenderArgs A)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;

  const int tile_x = blockIdx.x * 4 + wave;
  const int tile_y = blockIdx.y;
  if (tile_x * 8 >= A.W) return; // waves are independent: no barrier below
  const int col = tile_x * 8 + tx;
  const int lrow = tile_y * 8 + ty;
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow = (strip * A.nranks + A.rank) * A.strip_rows + (lrow0 - strip * A.strip_rows) + ty;
  const bool valid = (lrow < A.local_rows) && (grow < A.H) && (col < A.W);

  float accr = 0.0f, accg = 0.0f, accb = 0.0f;
  int out_id = -1;
  float out_t = MRT_INF;
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;
  const int nsamp = A.spp;
  const float DIVW = 13.0f, DIVH = 13.0f;
  const float v = float(grow) / float(A.H); // src/main.cu:137

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

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) { // pipeline/multisample_mesh.cuh:23-30
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
    float su = float(col) / float(A.W), sv = v; // src/main.cu:136
    if (nsamp > 1) { su = su + offx; sv = v + offy; }
    // current ray of the chain (pipeline/viewport.cuh:30-34 for depth 0)
    float ox = cam.ox, oy = cam.oy, oz = cam.oz;
    float dx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
    float dy = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
    float dz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;

    float stC[MAX_LEVELS][3], stL[MAX_LEVELS][3]; // per level: object colour, direct light
    bool stH[MAX_LEVELS];                         // ... and whether any light contributed
    int levels = 0;
    float inr = 0.0f, ing = 0.0f, inb = 0.0f;     // what the end of the chain returns
    bool alive = true;

    for (int depth = 0; depth <= A.max_depth; ++depth) {
      if (!__any(alive)) break;
      // ---- closest hit: src/main.cu:60-77, interval [0.01, inf] ----
      const float aa = dot3(dx, dy, dz, dx, dy, dz);
      float best_t = MRT_INF;
      int best_obj = 0x7fffffff, best_prim = -1;
      for (int i = 0; i < ns; ++i) { // core/sphere.cuh:32-73
        const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
        const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        const float b = dot3(ocx, ocy, ocz, dx, dy, dz);
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
        const float disc = b * b - aa * c;
        if (disc >= 0.0f) {
          float t;
          if (sphere_roots(aa, b, disc, 0.01f, MRT_INF, t)) {
            const int obj = sc.sph_obj[i];
            if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = i; }
          }
        }
      }
      for (int i = 0; i < nt; ++i) { // core/triangle.cuh:77-103
        const float *q = sc.tri_rec + 16 * size_t(i);
        const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
        const float rdn = dot3(dx, dy, dz, pnx, pny, pnz);
        const float t = (kp - dot3(ox, oy, oz, pnx, pny, pnz)) / rdn;
        const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
        const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
        const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
        const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
        const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
        if (hit) {
          const int obj = sc.tri_obj[i];
          if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = ns + i; }
        }
      }
      const bool have_hit = alive && best_prim >= 0;
      if (depth == 0) { // the ID / t buffers describe the primary hit
        out_id = have_hit ? best_obj : -1;
        out_t = best_t;
        if (have_hit && valid) n_hit++;
      }
      if (alive && !have_hit) { // background, src/main.cu:101-104: the chain ends in the sky
        float nx = dx, ny = dy, nz = dz;
        normalize3(nx, ny, nz);
        const float y = 0.5f * (ny + 1.0f);
        const float w0 = 1.0f - y;
        inr = w0 * 1.0f + y * 0.5f; ing = w0 * 1.0f + y * 0.7f; inb = w0 * 1.0f + y * 1.0f;
        levels = depth;
        alive = false;
      }

      // ---- hit record (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
      float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
      if (have_hit) {
        hpx = ox + best_t * dx; hpy = oy + best_t * dy; hpz = oz + best_t * dz;
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
      }

      // ---- shadow rays + direct light at this level: core/light.cuh:135-178 ----
      float lr = 0.0f, lg = 0.0f, lb = 0.0f;
      bool have_light = false;
      if (__any(have_hit)) {
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
          for (int i = 0; i < ns && __any(open); ++i) {
            const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
            const float ocx = hpx - cx, ocy = hpy - cy, ocz = hpz - cz;
            const float b = dot3(ocx, ocy, ocz, sdx, sdy, sdz);
            const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
            const float disc = b * b - sa * c;
            if (open && disc >= 0.0f) {
              float t;
              if (sphere_roots(sa, b, disc, 1e-3f, tl, t)) open = false;
            }
          }
          for (int i = 0; i < nt && __any(open); ++i) {
            const float *q = sc.tri_rec + 16 * size_t(i);
            const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
            const float rdn = dot3(sdx, sdy, sdz, pnx, pny, pnz);
            const float t = (kp - dot3(hpx, hpy, hpz, pnx, pny, pnz)) / rdn;
            const float px = hpx + t * sdx, py = hpy + t * sdy, pz = hpz + t * sdz;
            const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
            const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
            const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
            const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
            if (open && h) open = false;
          }
          if (need && valid) { n_shadow++; if (!open) n_occl++; }
          if (open) {
            // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168);
            // the viewing ray is the current ray of the chain
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

      // ---- push the level; continue with the bounced ray or end the chain ----
      if (have_hit) {
        const float *m = sc.mat_rgb + 3 * best_obj;
        stC[depth][0] = m[0]; stC[depth][1] = m[1]; stC[depth][2] = m[2];
        stL[depth][0] = lr; stL[depth][1] = lg; stL[depth][2] = lb;
        stH[depth] = have_light;
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
        if (depth + 1 > A.max_depth) { // what the recursive call returns beyond the limit, src/main.cu:51-53
          inr = 0.01f; ing = 0.01f; inb = 0.01f;
          levels = depth + 1;
          alive = false;
        } else {
          ox = hpx; oy = hpy; oz = hpz;
          dx = bdx; dy = bdy; dz = bdz;
        }
      }
    }

    // ---- fold the chain: src/main.cu:91-98, innermost level first ----
    float cr = inr, cg = ing, cb = inb;
    for (int k = levels - 1; k >= 0; --k) {
      const float mr = stC[k][0], mg = stC[k][1], mb = stC[k][2];
      if (stH[k]) { cr = cr * mr + stL[k][0] * mr; cg = cg * mg + stL[k][1] * mg; cb = cb * mb + stL[k][2] * mb; }
      else { cr = cr * mr; cg = cg * mg; cb = cb * mb; }
    }
    accr += cr; accg += cg; accb += cb; // src/main.cu:151
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
  }
}

hipError_t launch_render_bounce(const RenderArgs &A, hipStream_t stream)
{
  if (A.max_depth < 1 || A.max_depth > MAX_LEVELS - 1) return hipErrorInvalidValue;
  const int tiles_x = (A.W + 7) / 8, tiles_y = (A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  hipLaunchKernelGGL(render_bounce_kernel, dim3((tiles_x + 3) / 4, tiles_y), dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace mrt
