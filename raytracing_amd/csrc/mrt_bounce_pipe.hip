// mrt_bounce_pipe.hip -- bounce chains over scenes of at most 64 primitives, as a two-wave pipeline per tile.
// (config::MAX_RECURSION_DEPTH > 0: src/main.cu:26, :79-98; SURVEY 8(f).1-2.  mrt_bounce.hip has the semantics, the
// one-wave kernel this replaces for small scenes, and the kernel for larger ones.)
//
// Why two waves.  A pixel's 121 samples draw from ONE random stream one after the other (src/main.cu:135, :149-153),
// so a pixel is a sequential job: ~1050 chain steps where it sits between two mirrors.  A frame ends when its slowest
// wave does, and on gfx950 ONE wave alone on its SIMD issues a vector instruction only every 8-10 cycles whether the
// instructions depend on each other or not (tools/ubench_latency.hip: a dependent v_mul chain 10.0 cycles per op, four
// independent chains 7.5; with four waves on the SIMD 2.2 per op) -- instruction-level parallelism inside the wave buys
// nothing, the only way to shorten the job is to take instructions OFF the wave.  Per step, 12 000 cycles were
// (tools/timeline_cull.py REF, stamps in the loop of the 64 longest waves): closest hit 4 700, hit record 500, bounce
// + random draws 700 -- and shadow rays, shading, the push on the level stack and the fold of finished chains 6 000.
// Only the first group decides where the chain goes next; the second only turns what the chain met into colour.
//
// So a workgroup of two waves renders a tile, lane = pixel in both:
//   wave 0, the TRACER: primary rays, closest hits, hit records, bounces and random draws -- the chain itself;
//           per step it leaves one record per lane in LDS (hit point, normal, incoming direction, primitive, level);
//   wave 1, the SHADER: reads the records one step behind: shadow rays and direct light (core/light.cuh:135-178),
//           level stack, fold (src/main.cu:91-98), sky (src/main.cu:101-104), the pixel's sum over its samples.
// Nothing flows back from the shader to the tracer, so the two run side by side; one barrier per step and a ring of
// two record slots keep them at most one step apart.  Every operation on a pixel's data is the same fp32 operation
// in the same order as before (and as in the oracle): results are bit-identical.  Instructions per frame grow by the
// record traffic only (~3 %), and neither wave holds the other's state: no register spills any more.
#include "mrt_internal.h"
#include "mrt_device.h"
#include "mrt_cull_common.h"

#include <cmath>
#include <cstdlib>

namespace mrt {

namespace {
constexpr int PIPE_LEVELS = 9;  // depths 0..8
constexpr int REC_STRIDE = 24;  // floats per primitive record in LDS (16-byte aligned)
// a primitive's LDS record: sphere [0..3] centre, radius | triangle [0..15] the 16 floats; then for both
// [16] object number, [17] material kind, [18..20] material colour
enum { R_OBJ = 16, R_KIND = 17, R_RGB = 18 };
// a step's record, one per lane: [field][lane]
enum { F_FLAGS = 0, F_PRIM, F_HPX, F_HPY, F_HPZ, F_HNX, F_HNY, F_HNZ, F_DX, F_DY, F_DZ, F_WORDS };
enum : uint32_t { FL_ACT = 1u, FL_HIT = 2u, FL_LAST = 4u /* depth limit: the chain ends after this level */,
                  FL_DEPTH_SHIFT = 4, FL_DONE = 0x80000000u /* wave-uniform: no more records */ };
} // namespace

#ifdef MRT_TIMELINE
enum { TLP_WORDS = 16, TLP_MAX_WAVES = 1 << 17 };
__device__ unsigned long long g_timeline_pipe[TLP_WORDS * TLP_MAX_WAVES];
extern "C" int mrt_diag_timeline_pipe(unsigned long long *host_out, size_t words)
{
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline_pipe), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

template <bool MULTI>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8)))
void render_bounce_pipe_kernel(SmallArgs S)
{
  const RenderArgs &A = S.A;
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int role = threadIdx.x >> 6; // 0: tracer, 1: shader (wave-uniform)
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;

  const int tiles_x = (A.W + 7) >> 3;
  const int widx = blockIdx.x; // one tile per workgroup (launcher: grid = tiles)
  const int tile = A.tile_order ? int(A.tile_order[widx]) : widx; // heaviest tiles first (launcher), or row by row
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
#ifdef MRT_TIMELINE
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
  unsigned long long tl_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tl_last = tl_c0;
#define TLP_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tl_acc[k] += now_ - tl_last; tl_last = now_; } while (0)
#else
#define TLP_MARK(k) do { } while (0)
#endif
  const int col = tile_x * 8 + tx;
  const int lrow = tile_y * 8 + ty;
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid = (lrow < A.local_rows) && (grow < A.H) && (col < A.W);
  const int nsamp = MULTI ? A.spp : 1;

  __shared__ float scene_lds[64 * REC_STRIDE];
  __shared__ uint32_t ring[2][F_WORDS][64];
  __shared__ float offtab[32];

  // ---- can any primary ray of the tile hit anything?  (both waves: the same arithmetic, the same verdict) ----
  unsigned long long primary_mask; // primitives (bit = lane) a primary ray of the padded tile may hit
  {
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H));
    const float um = (float(tile_x * 8) + 3.5f) * rw, vm = (float(grow0) + 3.5f) * rh;
    float cax = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
    float cay = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
    float caz = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
    const float inv = fast_rsq(dot3(cax, cay, caz, cax, cay, caz));
    cax *= inv; cay *= inv; caz *= inv;
    float sinT = A.cone_ext * inv * (1.0f + 1e-4f) + 1e-6f;
    float cosT = fast_sqrt(fmaxf(1.0f - sinT * sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
    if (!(sinT < 0.99f) || !(cosT > 0.0f)) { cosT = -1.0f; sinT = 0.0f; } // cannot cull: every ball passes
    bool pass = false;
    if (lane < ns) { // lane = primitive (spheres first): cone test / screen-space edge functions, as in mrt_small.hip
      pass = ball_may_touch_cone(S.prm[9][lane], S.prm[10][lane], S.prm[11][lane], S.prm[12][lane], cax, cay, caz, cosT, sinT);
    } else if (lane < ns + nt) {
      bool cull = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float f = __builtin_fmaf(um, S.prm[3 * i + 1][lane], __builtin_fmaf(vm, S.prm[3 * i + 2][lane], S.prm[3 * i][lane]));
        cull |= (f < 0.0f); // NaN compares false: never culls
      }
      pass = !cull;
    }
    primary_mask = __ballot(pass);
  }
  const bool tile_has_candidates = primary_mask != 0;

  const float DIVW = 13.0f, DIVH = 13.0f;
  const float u0 = float(col) / float(A.W);  // src/main.cu:136
  const float v0 = float(grow) / float(A.H); // src/main.cu:137

  if (!tile_has_candidates) {
    // every chain is the sky colour of its primary ray: nothing is tested, nothing is drawn; the tracer does it alone
    if (role != 0) return;
    float accr = 0.0f, accg = 0.0f, accb = 0.0f;
    for (int sidx = 0; sidx < nsamp; ++sidx) {
      float su = u0, sv = v0;
      if (MULTI) { // pipeline/multisample_mesh.cuh:23-30
        const int r = sidx / 11, c = sidx - r * 11;
        su = u0 + (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
        sv = v0 + (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      }
      float nx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox; // pipeline/viewport.cuh:30-34
      float ny = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      float nz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
      normalize3(nx, ny, nz); // background, src/main.cu:101-104
      const float y = 0.5f * (ny + 1.0f);
      const float w0 = 1.0f - y;
      accr += w0 * 1.0f + y * 0.5f; accg += w0 * 1.0f + y * 0.7f; accb += w0 * 1.0f + y * 1.0f; // src/main.cu:151
    }
    if (valid) { // src/main.cu:155
      const float fn = float(nsamp);
      const size_t pix = size_t(A.frame_layout ? grow : lrow) * A.W + col;
      A.rgb[3 * pix + 0] = accr / fn; A.rgb[3 * pix + 1] = accg / fn; A.rgb[3 * pix + 2] = accb / fn;
      if (A.id) A.id[pix] = -1;
      if (A.t) A.t[pix] = MRT_INF;
    }
    if (A.stats) {
      unsigned long long *st = A.stats + 8u * (blockIdx.x & (unsigned)(A.stats_slots - 1));
      unsigned int x = (valid ? 1u : 0u) * (unsigned)nsamp;
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[0], (unsigned long long)x);
    }
    return;
  }

  // ---- the scene in LDS, one copy per workgroup, read by broadcast; written by the tracer's owning lanes ----
  const bool own_s = lane < ns, own_t = lane >= ns && lane < ns + nt;
  float pbx = 0.0f, pby = 0.0f, pbz = 0.0f, pbr = 0.0f; // the lane's primitive: sphere centre + radius / triangle bounding ball
  if (own_s) { pbx = sc.sph_cx[lane]; pby = sc.sph_cy[lane]; pbz = sc.sph_cz[lane]; pbr = sc.sph_r[lane]; }
  else if (own_t) { const int k = lane - ns; pbx = sc.tri_bx[k]; pby = sc.tri_by[k]; pbz = sc.tri_bz[k]; pbr = sc.tri_br[k]; }
  if (role == 0) {
    float *rec = &scene_lds[lane * REC_STRIDE];
    int obj = -1;
    if (own_s) {
      rec[0] = pbx; rec[1] = pby; rec[2] = pbz; rec[3] = pbr;
      obj = sc.sph_obj[lane];
    } else if (own_t) {
      const float4 *q4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(lane - ns));
#pragma unroll
      for (int k = 0; k < 4; ++k) reinterpret_cast<float4 *>(rec)[k] = q4[k];
      obj = sc.tri_obj[lane - ns];
    }
    if (obj >= 0) {
      rec[R_OBJ] = __int_as_float(obj);
      rec[R_KIND] = __int_as_float(int(sc.mat_kind[obj]));
      rec[R_RGB] = sc.mat_rgb[3 * obj]; rec[R_RGB + 1] = sc.mat_rgb[3 * obj + 1]; rec[R_RGB + 2] = sc.mat_rgb[3 * obj + 2];
    }
    if (MULTI) { // sample offsets of the 11 x 11 mesh (pipeline/multisample_mesh.cuh:23-30), tabulated once
      if (lane < 11) offtab[lane] = (-1.0f + 2.0f * float(lane + 1) / DIVW) / float(A.W);
      else if (lane >= 16 && lane < 27) offtab[lane] = (-1.0f + 2.0f * float(lane - 16 + 1) / DIVH) / float(A.H);
    }
  }
  __syncthreads();
  const unsigned long long all_prims = ns + nt >= 64 ? ~0ull : (1ull << (ns + nt)) - 1ull;

  if (role == 0) {
    // =========================================== the TRACER ====================================================
    float ox = cam.ox, oy = cam.oy, oz = cam.oz, dx = 0.0f, dy = 0.0f, dz = 0.0f; // current ray of the lane's chain
    auto primary_ray = [&](int sidx) { // pipeline/viewport.cuh:30-34
      float su = u0, sv = v0;
      if (MULTI) {
        const int r = sidx / 11, c = sidx - r * 11;
        su = u0 + offtab[c]; sv = v0 + offtab[16 + r];
      }
      ox = cam.ox; oy = cam.oy; oz = cam.oz;
      dx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
      dy = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      dz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
    };
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
    int out_id = -1;
    float out_t = MRT_INF;
    unsigned int n_hit = 0, n_steps = 0, n_iter = 0;
    int sidx = 0, depth = 0;
    bool work = valid; // lanes outside the frame have nothing to do
    if (work) primary_ray(0);
    // Two kinds of iteration (mrt_bounce.hip): primary -- every lane at level 0 steps against the tile's candidates,
    // lanes further down a chain wait; deep -- when no lane is at level 0, or deep_threshold lanes are waiting, the
    // waiting lanes step against every primitive.
    const int deep_threshold = A.s_cap > 0 ? A.s_cap : 64;
    int it = 0;
    while (true) {
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
      unsigned long long mask = primary_iter ? primary_mask : all_prims;
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float *rec = &scene_lds[j * REC_STRIDE];
        if (j < ns) { // core/sphere.cuh:32-73
          const float4 a = *reinterpret_cast<const float4 *>(rec);
          const int obj = __float_as_int(rec[R_OBJ]);
          const float ocx = ox - a.x, ocy = oy - a.y, ocz = oz - a.z;
          const float b = dot3(ocx, ocy, ocz, dx, dy, dz);
          const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - a.w * a.w;
          const float disc = b * b - aa * c;
          if (disc >= 0.0f) {
            float t;
            if (sphere_roots(aa, b, disc, 0.01f, MRT_INF, t)) {
              if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = j; }
            }
          }
        } else { // core/triangle.cuh:77-103
          const float4 qa = reinterpret_cast<const float4 *>(rec)[0];
          const float rdn = dot3(dx, dy, dz, qa.x, qa.y, qa.z);
          const float t = (qa.w - dot3(ox, oy, oz, qa.x, qa.y, qa.z)) / rdn;
          // the edge tests only matter in lanes whose t is in range and could still win: none in the wave, skip
          if (!__any(act & (rdn != 0.0f) & (0.01f <= t) & (t <= best_t))) continue;
          const float4 qb = reinterpret_cast<const float4 *>(rec)[1], qc = reinterpret_cast<const float4 *>(rec)[2],
                       qd = reinterpret_cast<const float4 *>(rec)[3];
          const int obj = __float_as_int(rec[R_OBJ]);
          const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
          const float e0 = dot3(qb.x, qb.y, qb.z, px, py, pz) - qb.w;
          const float e1 = dot3(qc.x, qc.y, qc.z, px, py, pz) - qc.w;
          const float e2 = dot3(qd.x, qd.y, qd.z, px, py, pz) - qd.w;
          const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
          if (hit) {
            if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = j; }
          }
        }
      }
      const bool have_hit = act && best_prim >= 0;
      if (act && depth == 0) { // the ID / t buffers describe the primary hit (of the last sample)
        out_id = have_hit ? best_obj : -1;
        out_t = best_t;
        if (have_hit) n_hit++;
      }
      // ---- hit record (core/sphere.cuh:45-47, core/triangle.cuh:96-97), from the winner's LDS record ----
      float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
      int kind = 1;
      if (have_hit) {
        hpx = ox + best_t * dx; hpy = oy + best_t * dy; hpz = oz + best_t * dz;
        const float *rec = &scene_lds[best_prim * REC_STRIDE];
        const float gx = rec[0], gy = rec[1], gz = rec[2]; // sphere centre / triangle plane normal
        kind = __float_as_int(rec[R_KIND]);
        if (best_prim < ns) {
          float nx = hpx - gx, ny = hpy - gy, nz = hpz - gz;
          normalize3(nx, ny, nz);
          hnx = nx; hny = ny; hnz = nz;
        } else {
          const bool front = dot3(dx, dy, dz, gx, gy, gz) < 0;
          hnx = front ? gx : -gx; hny = front ? gy : -gy; hnz = front ? gz : -gz;
        }
      }
      const bool last = have_hit && depth + 1 > A.max_depth; // the recursive call beyond the limit returns 0.01, src/main.cu:51-53
      // ---- the step's record for the shader ----
      {
        uint32_t(*slot)[64] = ring[it & 1];
        slot[F_FLAGS][lane] = (act ? FL_ACT : 0u) | (have_hit ? FL_HIT : 0u) | (last ? FL_LAST : 0u) | (uint32_t(depth) << FL_DEPTH_SHIFT);
        if (act) {
          slot[F_DX][lane] = __float_as_uint(dx); slot[F_DY][lane] = __float_as_uint(dy); slot[F_DZ][lane] = __float_as_uint(dz);
          if (have_hit) {
            slot[F_PRIM][lane] = uint32_t(best_prim);
            slot[F_HPX][lane] = __float_as_uint(hpx); slot[F_HPY][lane] = __float_as_uint(hpy); slot[F_HPZ][lane] = __float_as_uint(hpz);
            slot[F_HNX][lane] = __float_as_uint(hnx); slot[F_HNY][lane] = __float_as_uint(hny); slot[F_HNZ][lane] = __float_as_uint(hnz);
          }
        }
      }
      __syncthreads(); // the shader may read this slot now; the other slot is free (its reader has passed the barrier before)
      ++it;
      // ---- the bounce (src/main.cu:80-83) happens at every level, the last one included: a diffuse surface
      //      advances the pixel's stream even when its bounce ray is never followed ----
      bool finish = act && !have_hit; // the chain ends in the sky
      if (have_hit) {
        float bdx, bdy, bdz;
        if (use_rng && kind == 0) { // DiffuseMaterial::bounce, core/material.cuh:52-61
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
        if (last) finish = true;
        else {
          ox = hpx; oy = hpy; oz = hpz;
          dx = bdx; dy = bdy; dz = bdz;
          depth++;
        }
      }
      if (finish) { // next sample of the pixel
        sidx++;
        depth = 0;
        work = sidx < nsamp;
        if (work) primary_ray(sidx);
      }
    }
    ring[it & 1][F_FLAGS][lane] = FL_DONE;
    __syncthreads();

    if (valid) {
      const size_t pix = size_t(A.frame_layout ? grow : lrow) * A.W + col;
      if (A.id) A.id[pix] = out_id;
      if (A.t) A.t[pix] = out_t;
    }
#ifdef MRT_TIMELINE
    if (lane == 0 && tile < TLP_MAX_WAVES) {
      unsigned long long *tlp = g_timeline_pipe + (size_t)TLP_WORDS * tile;
      tlp[0] = tl_rt0; tlp[1] = __builtin_amdgcn_s_memrealtime(); tlp[2] = __builtin_amdgcn_s_memtime() - tl_c0;
      tlp[3] = (unsigned long long)n_steps | ((unsigned long long)n_iter << 32);
    }
#endif
    if (A.stats) {
      // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
      unsigned long long *st = A.stats + 8u * (blockIdx.x & (unsigned)(A.stats_slots - 1));
      unsigned int v2[2] = {(valid ? 1u : 0u) * (unsigned)nsamp, n_hit};
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        unsigned int x = v2[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0 && x) atomicAdd(&st[k == 0 ? 0 : 2], (unsigned long long)x);
      }
      if (lane == 0 && n_iter) { // reported as exact_tests / clusters_opened: chain steps of all lanes / 64 x loop iterations
        atomicAdd(&st[4], (unsigned long long)n_steps);
        atomicAdd(&st[5], (unsigned long long)n_iter * 64ull);
      }
    }
    return;
  }

  // ============================================= the SHADER ====================================================
  float accr = 0.0f, accg = 0.0f, accb = 0.0f;
  unsigned int n_shadow = 0, n_occl = 0;
  float stC[PIPE_LEVELS][3], stL[PIPE_LEVELS][3]; // per level: object colour, direct light (registers: static indices only)
  unsigned int stH = 0;                           // bit k: a light contributed at level k
  int it = 0;
  while (true) {
    __syncthreads(); // the tracer has written slot it & 1
    uint32_t(*slot)[64] = ring[it & 1];
    ++it;
    const uint32_t fl = slot[F_FLAGS][lane];
    if (fl & FL_DONE) break; // written by every lane of the tracer: wave-uniform
    const bool act = fl & FL_ACT, have_hit = fl & FL_HIT, last = fl & FL_LAST;
    const int depth = int((fl >> FL_DEPTH_SHIFT) & 15u);
    float dx = 0.0f, dy = 0.0f, dz = 0.0f;
    if (act) { dx = __uint_as_float(slot[F_DX][lane]); dy = __uint_as_float(slot[F_DY][lane]); dz = __uint_as_float(slot[F_DZ][lane]); }
    float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
    int prim = 0;
    if (have_hit) {
      prim = int(slot[F_PRIM][lane]);
      hpx = __uint_as_float(slot[F_HPX][lane]); hpy = __uint_as_float(slot[F_HPY][lane]); hpz = __uint_as_float(slot[F_HPZ][lane]);
      hnx = __uint_as_float(slot[F_HNX][lane]); hny = __uint_as_float(slot[F_HNY][lane]); hnz = __uint_as_float(slot[F_HNZ][lane]);
    }
    TLP_MARK(0);
    bool finish = false;
    int levels = 0;
    float inr = 0.0f, ing = 0.0f, inb = 0.0f; // what the end of the chain returns
    if (act && !have_hit) {                   // the chain ends in the sky: background, src/main.cu:101-104
      float nx = dx, ny = dy, nz = dz;
      normalize3(nx, ny, nz);
      const float y = 0.5f * (ny + 1.0f);
      const float w0 = 1.0f - y;
      inr = w0 * 1.0f + y * 0.5f; ing = w0 * 1.0f + y * 0.7f; inb = w0 * 1.0f + y * 1.0f;
      levels = depth;
      finish = true;
    }
    TLP_MARK(1);

    // ---- shadow rays + direct light at this level: core/light.cuh:135-178 ----
    float lr = 0.0f, lg = 0.0f, lb = 0.0f;
    bool have_light = false;
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f; // the hit object's colour
    if (__any(have_hit)) {
      bool diffuse_obj = false;
      if (have_hit) {
        const float *rec = &scene_lds[prim * REC_STRIDE];
        diffuse_obj = __float_as_int(rec[R_KIND]) == 0;
        m0 = rec[R_RGB]; m1 = rec[R_RGB + 1]; m2 = rec[R_RGB + 2];
      }
      // ball around the hit points of this step (centre = the first one); with a light it spans the bundle of shadow
      // rays that each primitive's bound is tested against (mrt_cull_common.h), one primitive per lane
      Bundle B;
      {
        const int first = __builtin_ctzll(__ballot(have_hit));
        B.mx = bcast(hpx, first); B.my = bcast(hpy, first); B.mz = bcast(hpz, first);
        const float bex = hpx - B.mx, bey = hpy - B.my, bez = hpz - B.mz;
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
          const float *rec = &scene_lds[j * REC_STRIDE];
          if (j < ns) {
            const float4 a = *reinterpret_cast<const float4 *>(rec);
            const float ocx = hpx - a.x, ocy = hpy - a.y, ocz = hpz - a.z;
            const float b = dot3(ocx, ocy, ocz, sdx, sdy, sdz);
            const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - a.w * a.w;
            const float disc = b * b - sa * c;
            if (open && disc >= 0.0f) {
              float t;
              if (sphere_roots(sa, b, disc, 1e-3f, tl, t)) open = false;
            }
          } else {
            const float4 qa = reinterpret_cast<const float4 *>(rec)[0];
            const float rdn = dot3(sdx, sdy, sdz, qa.x, qa.y, qa.z);
            const float t = (qa.w - dot3(hpx, hpy, hpz, qa.x, qa.y, qa.z)) / rdn;
            if (!__any(open & (rdn != 0.0f) & (1e-3f <= t) & (t <= tl))) continue;
            const float4 qb = reinterpret_cast<const float4 *>(rec)[1], qc = reinterpret_cast<const float4 *>(rec)[2],
                         qd = reinterpret_cast<const float4 *>(rec)[3];
            const float px = hpx + t * sdx, py = hpy + t * sdy, pz = hpz + t * sdz;
            const float e0 = dot3(qb.x, qb.y, qb.z, px, py, pz) - qb.w;
            const float e1 = dot3(qc.x, qc.y, qc.z, px, py, pz) - qc.w;
            const float e2 = dot3(qd.x, qd.y, qd.z, px, py, pz) - qd.w;
            const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
            if (open && h) open = false;
          }
        }
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
    TLP_MARK(2);

    // ---- push the level ----
    if (have_hit) {
#pragma unroll
      for (int k = 0; k < PIPE_LEVELS; ++k)
        if (k == depth) {
          stC[k][0] = m0; stC[k][1] = m1; stC[k][2] = m2;
          stL[k][0] = lr; stL[k][1] = lg; stL[k][2] = lb;
        }
      stH = have_light ? (stH | (1u << depth)) : (stH & ~(1u << depth));
      if (last) { // what the recursive call returns beyond the limit, src/main.cu:51-53
        inr = 0.01f; ing = 0.01f; inb = 0.01f;
        levels = depth + 1;
        finish = true;
      }
    }
    TLP_MARK(3);
    // ---- a chain ended: fold it (src/main.cu:91-98, innermost level first) and add it to the pixel ----
    if (__any(finish)) {
      float cr = inr, cg = ing, cb = inb;
#pragma unroll
      for (int k = PIPE_LEVELS - 1; k >= 0; --k) {
        if (!__any(finish && k < levels)) continue; // most chains are short: skip the levels no lane has
        if (k < levels) {
          const float mr = stC[k][0], mg = stC[k][1], mb = stC[k][2];
          if (stH & (1u << k)) { cr = cr * mr + stL[k][0] * mr; cg = cg * mg + stL[k][1] * mg; cb = cb * mb + stL[k][2] * mb; }
          else { cr = cr * mr; cg = cg * mg; cb = cb * mb; }
        }
      }
      if (finish) { accr += cr; accg += cg; accb += cb; } // src/main.cu:151
    }
    TLP_MARK(4);
  }

  if (valid) { // src/main.cu:155
    const float fn = float(nsamp);
    const size_t pix = size_t(A.frame_layout ? grow : lrow) * A.W + col;
    A.rgb[3 * pix + 0] = accr / fn;
    A.rgb[3 * pix + 1] = accg / fn;
    A.rgb[3 * pix + 2] = accb / fn;
  }
#ifdef MRT_TIMELINE
  if (lane == 0 && tile < TLP_MAX_WAVES) {
    unsigned long long *tlp = g_timeline_pipe + (size_t)TLP_WORDS * tile;
    for (int k = 0; k < 8; ++k) tlp[4 + k] = tl_acc[k];
    tlp[12] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  if (A.stats) {
    unsigned long long *st = A.stats + 8u * (blockIdx.x & (unsigned)(A.stats_slots - 1));
    unsigned int v2[2] = {n_shadow, n_occl};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      unsigned int x = v2[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[k == 0 ? 1 : 3], (unsigned long long)x);
    }
  }
}

// S: the small-scene arguments (small_prepare_args, P = 1) with s_cap = the deep-iteration threshold and tile_order set
// or null (mrt_bounce.hip's launcher, which also orders the tiles by predicted cost)
hipError_t launch_render_bounce_pipe(const SmallArgs &S, hipStream_t stream)
{
  const int tiles_x = (S.A.W + 7) / 8, tiles_y = (S.A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  const dim3 grid(tiles_x * tiles_y), block(128);
  if (S.A.spp > 1) hipLaunchKernelGGL(render_bounce_pipe_kernel<true>, grid, block, 0, stream, S);
  else hipLaunchKernelGGL(render_bounce_pipe_kernel<false>, grid, block, 0, stream, S);
  return hipGetLastError();
}

} // namespace mrt
