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
//   * primary rays: a sphere is bounded by its SCREEN RECTANGLE (the tile centres whose padded tile can hold a ray
//     that touches it: four compares per tile; small_primary_functions), shadow rays by the swept-ball tests of
//     mrt_cull_common.h; triangles by SCREEN-SPACE EDGE FUNCTIONS.  For a triangle (plane n, k; edge planes m_i, c_i) and a ray o + t d accepted by the
//     exact test (t > 0), e_i(p) = (w_i . d) / (n . d) with w_i = (m_i.o - c_i) n + (k - n.o) m_i, and
//     sign(n.d) = sign(k - n.o); w_i . d is affine in the pixel coordinates, so its maximum over the tile is
//     its value at the tile centre plus a constant.  If that maximum is below a margin for one edge, no ray
//     of the tile can pass that edge test and the triangle is skipped.  The functions depend on the camera:
//     the launcher computes them in double for every render (small_primary_functions) and they travel in the
//     kernel arguments.  The same construction bounds shadow rays: e_i at the point where the shadow ray of
//     hit point p meets the plane is affine in p, for directional and for point lights, and is bounded over
//     the ball around the tile's hit points; these functions depend on scene and lights only and are built
//     at upload (mrt_scene_build.h, shadow_functions);
//   * a wave is one workgroup; rows of tiles are launched starting where the view's geometry begins (small_first_row_fast,
//     on the host, per new view), and a tile no ray of which can hit anything takes a sky-only path;
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
#include <vector>

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

#ifndef MRT_TIMELINE
#define TL_MARK(k) do { } while (0)
#endif
// Ablation builds (make variant DEFS=-DMRT_ABL=n; never the product library): bit 0 every tile takes the sky path,
// bit 1 nothing is stored (the stores stay in the code behind a condition that is never true), bit 2 waves end
// after the culling prologue, bit 3 approximate sqrt / division in the sky colour
#ifndef MRT_ABL
#define MRT_ABL 0
#endif
// Phase cuts (make variant DEFS=-DMRT_CUT=n; never the product library): tiles with candidates end after phase n --
// 8 rays set up, 9 closest hits, 10 hit records, 11 shadow bundle, 12 shadow candidates, 13 shadow rays, 14 shading --
// without storing; what the phase computed feeds a condition that is never true, so it is not optimised away.  The
// counters of two neighbouring cuts differ by one phase (tools/phase_counters.sh, profiles/*_phase_counters.json).
#ifndef MRT_CUT
#define MRT_CUT 0
#endif
#define MRT_CUT_AT(k, expr) do { if (MRT_CUT == (k)) { if ((expr) == 1234.5f) A.rgb[0] = 1.0f; return; } } while (0)
#ifndef MRT_SMALL_BANDS_LOG2
#define MRT_SMALL_BANDS_LOG2 0 // 3: rows of tiles launched interleaved from 8 bands of the frame (measured: slower)
#endif
#ifndef MRT_SMALL_NT
#define MRT_SMALL_NT 1 // frame stores are non-temporal
#endif
template <class T> __device__ __forceinline__ void frame_store(const T &x, T *p)
{
  if (MRT_SMALL_NT) __builtin_nontemporal_store(x, p); else *p = x;
}
#ifdef MRT_SMALL_ORDER_TABLE
__device__ unsigned short g_order[8192];
#endif
#ifdef MRT_TIMELINE
// Diagnostic build only (tools/timeline.py): per wave, shader-clock stamps at entry, after the culling prologue,
// before the stores and at exit, the 100 MHz real-time clock at entry, and where the wave ran.
enum { TL_WORDS = 16, TL_MAX_WAVES = 1 << 16 };
#define TL_MARK(k) do { if (tl_on && lane == 0) tlp[k] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_timeline[TL_WORDS * TL_MAX_WAVES];
#endif

// SmallArgs (mrt_internal.h): the common kernel arguments plus prm[field][lane], read straight from the argument
// segment, one memory round trip after launch.
//
// SmallHot: what a wave needs before anything has arrived from memory.  These are the FIRST kernel arguments,
// 5 dwords, and gfx950 loads them into SGPRs while it launches the wave (kernarg preload; Makefile:
// -mllvm -amdgpu-kernarg-preload-count=5 for this file): the wave can issue the loads of its primitive's culling
// parameters and of its u / v table entries in its first cycles instead of after a scalar round trip (measured:
// the prologue was 3000 of a sky tile's 6200 cycles).
struct SmallHot {
  uint32_t wh;       // (W - 1) | (H - 1) << 16
  float part_flag;   // negative: the frame is strip-partitioned (rows need the strip rule, which reads SmallArgs)
  uint32_t meta;     // ns | nt << 8 | rows of tiles per band << 16 (row interleave, below)
  const float *utab; // u = col / W table; the v = row / H table follows at utab + uv_table_stride(W)
};
static_assert(sizeof(SmallHot) == 6 * 4, "the preloaded arguments are 5 dwords (+ 1 of alignment)");

template <int P, bool MULTI, int TPW, int LX>
__device__ __forceinline__ void render_small_body(uint32_t wh, float cone_ext_s,
                                                  uint32_t meta, const float *utab_hot, const SmallArgs &S)
{
  const RenderArgs &A = S.A;
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x; // one wave per workgroup
  constexpr int LY = 64 / LX;   // a tile is LX lanes wide and LY rows high: 8 x 8 lanes or 16 x 4
  const int tx = lane & (LX - 1), ty = lane / LX;
  constexpr int TW = LX * P;    // tile width in pixels
  // ---- from the preloaded arguments only: no memory access has returned yet ----
  const int W = int(wh & 0xffffu) + 1, H = int(wh >> 16) + 1;
  constexpr int tpw = TPW;      // tiles per wave
  const int ns = int(meta & 0xffu), nt = int((meta >> 8) & 0xffu); // ns + nt <= 64
  // Experiment knob (MRT_SMALL_BANDS_LOG2 > 0, default 0 = row order): the grid's row y becomes row
  // (y % BANDS) * band_rows + y / BANDS of the frame, so consecutive workgroups come from BANDS places spread over its
  // height and sky tiles (bound by the stores) run beside hit tiles (bound by instruction issue).  Measured on the
  // default frame: SLOWER, 17.8 -> 20.1 us with 2 or 8 bands (gpurun_out/exp_bands_*.txt), as is raising the priority
  // of hit waves (19.9 us): vector loads of a hit wave queue behind the stores of the sky waves that share its CU.
  [[maybe_unused]] constexpr int BANDS = 1 << MRT_SMALL_BANDS_LOG2;
  const int band_rows = int(meta >> 16);
#ifdef MRT_SMALL_ORDER_TABLE
  // experiment builds only (tools/exp_order.py): the row of tiles comes from a table set through mrt_diag_set_order
  int tile_y = g_order[blockIdx.y];
  (void)band_rows;
#elif MRT_SMALL_BANDS_LOG2 == 0
  // Rows of tiles are launched starting at row `band_rows` (here: the launcher's row0, small_first_row) and wrap
  // around: tiles with candidates live four times as long as sky tiles, so the band of the frame that holds the
  // geometry should start first and the sky rows fill the machine while its last waves drain.
  int tile_y = int(blockIdx.y) + band_rows;
  tile_y = tile_y >= int(gridDim.y) ? tile_y - int(gridDim.y) : tile_y;
#else
  const int tile_y = int(blockIdx.y & (BANDS - 1)) * band_rows + int(blockIdx.y >> MRT_SMALL_BANDS_LOG2);
#endif
  const int tile_x0 = blockIdx.x * tpw;
  const int lrow0 = tile_y * LY, lrow = lrow0 + ty;
  if (lrow0 >= H) return; // the grid is padded to BANDS * band_rows rows (and no rank has more rows than the frame)
#ifdef MRT_TIMELINE
  const unsigned tl_idx = blockIdx.y * gridDim.x + blockIdx.x;
  const bool tl_on = tl_idx < TL_MAX_WAVES;
  unsigned long long *tlp = g_timeline + (size_t)TL_WORDS * (tl_on ? tl_idx : 0);
  if (tl_on && lane == 0) {
    tlp[0] = __builtin_amdgcn_s_memtime();
    tlp[4] = __builtin_amdgcn_s_memrealtime();
    tlp[5] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |               // HW_REG_HW_ID
            ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) & 0xF) << 32); // HW_REG_XCC_ID
  }
#endif
  // the primitive this lane owns (spheres first, then triangles): culling parameters, straight from the argument
  // segment (its address is in SGPRs from the start)
  const bool own_s = lane < ns, own_t = (lane >= ns) && (lane < ns + nt);
  // fields 0..11 of the lane's column, whatever the lane owns (a sphere uses 0..3: its screen rectangle; a triangle
  // all twelve: four edge functions; lanes that own nothing read their column too and ignore it): no branches, and
  // nothing to initialise
  float cp[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) cp[i] = S.prm[i][lane];
  // rows: without a strip partition local row = frame row; with one, the strip rule (needs arguments from memory)
  int grow0 = lrow0;
  if (cone_ext_s < 0.0f) {
    if (lrow0 >= A.local_rows) return;
    const int strip = lrow0 / A.strip_rows; // strips are multiples of 8 rows, LY divides 8: a tile never straddles two
    grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  }
  const int grow = grow0 + ty;
  // u, v of the lane's pixels: tables hold float(i) / float(W) and float(i) / float(H) (uv_table_kernel); v is the
  // same for all of the wave's tiles, u is fetched one tile ahead
  const float *vtab_hot = utab_hot + uv_table_stride(W);
#ifndef MRT_SMALL_UV
#define MRT_SMALL_UV 0 // experiment: 1 = u, v by IEEE division in the wave instead of from the tables
#endif
#if MRT_SMALL_UV
  const float v = float(grow) / float(H); // src/main.cu:137
#else
  const float v = vtab_hot[grow < H ? grow : H - 1]; // rows beyond the frame are never written
#endif
  using UV = typename VecOf<P>::f;

  // ---- cones around the tiles' rays, each padded by one pixel on every side (covers the 121-spp offsets): every
  //      ray of a padded tile is its centre ray plus an offset of norm <= cone_ext (launcher); which primitives
  //      can any ray of tile j hit? ----
  const float rw = fast_rcp(float(W)), rh = fast_rcp(float(H)); // bound arithmetic: approximate is fine
  const float vm = (float(grow0) + 0.5f * float(LY - 1)) * rh;
  unsigned long long pmask[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    pmask[j] = 0ull;
    { // wave-uniform
      const float um = (float((tile_x0 + j) * TW) + 0.5f * float(TW - 1)) * rw;
      bool pass = false;
      if (own_s) {
        // the sphere's screen rectangle of tile centres (launcher: small_primary_functions); NaN never culls
        pass = !(um < cp[0]) & !(um > cp[1]) & !(vm < cp[2]) & !(vm > cp[3]);
      } else if (own_t) {
        bool cull = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = __builtin_fmaf(um, cp[3 * i + 1], __builtin_fmaf(vm, cp[3 * i + 2], cp[3 * i]));
          cull |= (f < 0.0f); // NaN compares false: never culls
        }
        pass = !cull;
      }
      pmask[j] = __ballot(pass);
    }
  }
#ifdef MRT_TIMELINE
  if (tl_on && lane == 0) {
    tlp[1] = __builtin_amdgcn_s_memtime();
    unsigned long long any = 0;
    for (int j = 0; j < TPW; ++j) any |= pmask[j];
    tlp[7] = any;
  }
#endif

  const unsigned long long smask_all = ns >= 64 ? ~0ull : ((1ull << ns) - 1ull);
  const bool valid_row = (lrow < A.local_rows) && (grow < A.H);
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0, n_valid = 0;
  unsigned int n_exact = 0, n_clusters = 0; // wave-uniform diagnostics (lane 0 reports them)
  const int nsamp = MULTI ? A.spp : 1; // MULTI = false: one sample, no accumulation across samples
  const float DIVW = 13.0f, DIVH = 13.0f;

  // ================= one tile after the other =================
  auto render_tile = [&](const int j) {
  const int tile_x = tile_x0 + j;
  if (tile_x * TW >= A.W) return;
  unsigned long long primary_mask = pmask[0];
#pragma unroll
  for (int k = 1; k < TPW; ++k) primary_mask = j == k ? pmask[k] : primary_mask;
  const int col0 = tile_x * TW + tx * P;
  float u0[P];
#if MRT_SMALL_UV
#pragma unroll
  for (int p = 0; p < P; ++p) u0[p] = float(col0 + p) / float(W); // src/main.cu:136
#else
  {
    const UV uu = *reinterpret_cast<const UV *>(utab_hot + col0); // the table is padded to whole tiles
    const float *up = reinterpret_cast<const float *>(&uu);
#pragma unroll
    for (int p = 0; p < P; ++p) u0[p] = up[p];
  }
#endif

  float accr[P], accg[P], accb[P];
  int out_id[P];
  float out_t[P];
  bool valid[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    accr[p] = 0.0f; accg[p] = 0.0f; accb[p] = 0.0f;
    out_id[p] = -1; out_t[p] = MRT_INF;
    valid[p] = valid_row && (col0 + p < A.W);
    n_valid += valid[p] ? 1u : 0u;
  }

  // sky colour of a ray that hits nothing: src/main.cu:101-104
  auto sky_colour = [](float nx, float ny, float nz, float &cr, float &cg, float &cb) {
    if (MRT_ABL & 8) ny = ny * fast_rsq(dot3(nx, ny, nz, nx, ny, nz)); else
    normalize3(nx, ny, nz);
    const float y = 0.5f * (ny + 1.0f);
    const float w0 = 1.0f - y;
    cr = w0 * 1.0f + y * 0.5f; cg = w0 * 1.0f + y * 0.7f; cb = w0 * 1.0f + y * 1.0f;
  };
  // sub-pixel offset of sample sidx: pipeline/multisample_mesh.cuh:23-30
  auto sample_offset = [&](int sidx, float &offx, float &offy) {
    offx = 0.0f; offy = 0.0f;
    if (nsamp > 1) {
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
  };

  if (MRT_ABL & 4) { if (W != 12345) return; }
  if ((MRT_ABL & 1) || primary_mask == 0ull) {
    // ---- no ray of this (padded) tile can hit a primitive: every sample is sky.  Same expressions as below for the
    //      ray and the colour; none of the hit machinery, none of the lane's primitive data (75 % of the default
    //      frame's pixels) ----
    for (int sidx = 0; sidx < nsamp; ++sidx) {
      float offx, offy;
      sample_offset(sidx, offx, offy);
      const float sv = nsamp > 1 ? v + offy : v;
      n_clusters++;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const float su = nsamp > 1 ? u0[p] + offx : u0[p];
        const float dx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox; // pipeline/viewport.cuh:30-34
        const float dy = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
        const float dz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
        float cr, cg, cb;
        sky_colour(dx, dy, dz, cr, cg, cb);
        accr[p] += cr; accg[p] += cg; accb[p] += cb; // src/main.cu:151
      }
    }
  } else {
  // ---- the rest of what the lane keeps about its primitive: the ray-independent terms of the exact primary
  //      tests, its bound and its centre / plane normal (shadow rays, hit records).  Fetched here, in tiles with
  //      candidates only: 70 % of the default frame's tiles are sky and never need it ----
  float bx = 0.0f, by = 0.0f, bz = 0.0f, brad = 0.0f; // sphere centre + radius / triangle bound ball
  float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;               // sphere centre / triangle plane normal
  float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, e3 = 0.0f;    // sphere: o - c, |o - c|^2 - r^2;  triangle: e0 = k - n.o
  float rr = 0.0f;                                     // sphere: r^2
  int pobj = 0;
  if (own_s) {
    e0 = S.prm[4][lane]; e1 = S.prm[5][lane]; e2 = S.prm[6][lane]; e3 = S.prm[7][lane];
    pobj = __float_as_int(S.prm[8][lane]);
    bx = sc.sph_cx[lane]; by = sc.sph_cy[lane]; bz = sc.sph_cz[lane]; brad = sc.sph_r[lane];
    g0 = bx; g1 = by; g2 = bz;
    rr = brad * brad;
  } else if (own_t) {
    const int k = lane - ns;
    e0 = S.prm[12][lane];
    pobj = __float_as_int(S.prm[13][lane]);
    bx = sc.tri_bx[k]; by = sc.tri_by[k]; bz = sc.tri_bz[k]; brad = sc.tri_br[k];
    g0 = sc.tri[k]; g1 = sc.tri[nt + k]; g2 = sc.tri[2 * nt + k];
  }
  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx, offy;
    sample_offset(sidx, offx, offy);
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

    TL_MARK(8); // rays set up
    MRT_CUT_AT(8, dx[0] + dy[P - 1] + dz[0] + aa[P - 1]);
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

    TL_MARK(9); // closest hits found
    MRT_CUT_AT(9, best_t[0] + best_t[P - 1] + float(best_obj[0] + best_k[P - 1]));
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

    TL_MARK(10); // hit records
    MRT_CUT_AT(10, hpx[0] + hpy[P - 1] + hpz[0] + hnx[P - 1] + hny[0] + hnz[P - 1]);
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
      TL_MARK(11); // bundle
      MRT_CUT_AT(11, B.rho + reff_shadow + Pmax + B.mx);
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
          TL_MARK(12); // shadow candidates known (last light)
          MRT_CUT_AT(12, float(shadow_mask & 0xffffffull));
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

        TL_MARK(13); // shadow rays traced (last light)
        MRT_CUT_AT(13, float(open[0]) + float(open[P - 1]) + float(need[0]));
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

    TL_MARK(14); // lights done
    if (MRT_CUT >= 12 && MRT_CUT <= 14) { if (lr[0] + lg[P - 1] + lb[0] + float(have_light[P - 1]) == 1234.5f) A.rgb[0] = 1.0f; return; }
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
        sky_colour(dx[p], dy[p], dz[p], cr, cg, cb);
      }
      accr[p] += cr; accg[p] += cg; accb[p] += cb; // src/main.cu:151
      out_id[p] = have_hit[p] ? best_obj[p] : -1;
      out_t[p] = best_t[p];
    }
  }
  }

  // ---- framebuffer: src/main.cu:155 (x / 1.0f == x, so the division is skipped for one sample) ----
  if (nsamp > 1) {
    const float fn = float(nsamp);
#pragma unroll
    for (int p = 0; p < P; ++p) { accr[p] = accr[p] / fn; accg[p] = accg[p] / fn; accb[p] = accb[p] / fn; }
  }
  const size_t pix0 = size_t(A.frame_layout ? grow : lrow) * A.W + col0;
#ifdef MRT_TIMELINE
  if (tl_on && lane == 0) tlp[2] = __builtin_amdgcn_s_memtime();
#endif
  if ((MRT_ABL & 2) && accr[0] != 1234.5f) return;
#ifndef MRT_SMALL_LDS_ROWS
#define MRT_SMALL_LDS_ROWS 0
#endif
  // Experiment (make variant DEFS=-DMRT_SMALL_LDS_ROWS=1; round 3, OFF): whole tiles of the two-pixel instantiation write
  // their RGB rows THROUGH LDS.  A lane's 24 bytes are three dwordx2 at a 24-byte stride, so each of the three store
  // instructions touches every 64-byte segment of the tile's eight 192-byte rows but fills a third of it; transposed
  // through LDS the rows go out as 96 pieces of 16 bytes, four lanes to a segment (two instructions).  In a store-only
  // kernel of this tile shape (tools/ubench_launch.hip) that is worth 2.8 us: 10.2 us as the kernel stores
  // (non-temporal; 8.1 plain; dwordx4 + dwordx2 or two dwordx3 per lane: no change) against 7.4 us through LDS, which is
  // the write roof of a frame of this size (one contiguous 41.5 MB fill: 7.2-7.9 us = 5.2-5.8 TB/s).  In THIS kernel it
  // LOSES 0.8 us (default frame 16.65 against 15.86 us, HIP events; with plain stores 17.2): the LDS round trip and the
  // wave barrier sit at the end of every wave's life, and the frame is bound by how long its 16 200 waves live, not by
  // how well their stores combine.
  constexpr bool LDS_ROWS = MRT_SMALL_LDS_ROWS && P == 2 && LX == 8 && !MULTI;
  bool lds_rows = false;
  if constexpr (LDS_ROWS) lds_rows = A.wide && (((uintptr_t)A.rgb & 15) == 0) && ((W & 3) == 0) && __all(valid_row && col0 + P <= A.W);
  if (LDS_ROWS && lds_rows) {
    if constexpr (LDS_ROWS) {
      __shared__ float rgb_rows[8 * 48];
      float *mine = &rgb_rows[ty * 48 + tx * 6];
      *reinterpret_cast<float2 *>(mine) = make_float2(accr[0], accg[0]);
      *reinterpret_cast<float2 *>(mine + 2) = make_float2(accb[0], accr[1]);
      *reinterpret_cast<float2 *>(mine + 4) = make_float2(accg[1], accb[1]);
      __builtin_amdgcn_wave_barrier(); // written and read by this wave only
      const size_t row_base = size_t(A.frame_layout ? grow0 : lrow0); // the tile's first row in the buffers
      const int tile_col0 = col0 - tx * P;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int piece = h * 64 + lane; // 96 pieces of 16 bytes, 12 to a row
        if (piece < 96) {
          const int r = piece / 12, c = piece - r * 12;
          const nfloat4 val = *reinterpret_cast<const nfloat4 *>(&rgb_rows[r * 48 + c * 4]);
          frame_store(val, reinterpret_cast<nfloat4 *>(A.rgb + 3 * ((row_base + r) * A.W + tile_col0) + c * 4));
        }
      }
      using NF = typename VecOf<P>::nf;
      using NI = typename VecOf<P>::ni;
      if (A.id) frame_store(*reinterpret_cast<const NI *>(&out_id[0]), reinterpret_cast<NI *>(A.id + pix0));
      if (A.t) frame_store(*reinterpret_cast<const NF *>(&out_t[0]), reinterpret_cast<NF *>(A.t + pix0));
      if constexpr (TPW > 1) __builtin_amdgcn_wave_barrier(); // the next tile of the wave writes the rows again
    }
  } else if (A.wide && valid_row && col0 + P <= A.W) {
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
    for (int k = 0; k < 3; ++k) frame_store(*reinterpret_cast<const NF *>(&o[P * k]), reinterpret_cast<NF *>(dst) + k);
    if (A.id) frame_store(*reinterpret_cast<const NI *>(&out_id[0]), reinterpret_cast<NI *>(A.id + pix0));
    if (A.t) frame_store(*reinterpret_cast<const NF *>(&out_t[0]), reinterpret_cast<NF *>(A.t + pix0));
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
  }; // render_tile
  if constexpr (TPW == 1) {
    render_tile(0);
  } else {
#pragma unroll 1
    for (int j = 0; j < TPW; ++j) render_tile(j);
  }

#ifdef MRT_TIMELINE
  if (tl_on && lane == 0) { tlp[3] = __builtin_amdgcn_s_memtime(); tlp[6] = __builtin_amdgcn_s_memrealtime(); }
#endif
  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * ((blockIdx.x + 7u * blockIdx.y) & (unsigned)(A.stats_slots - 1));
    unsigned int v4[4] = {n_valid * (unsigned)nsamp, n_shadow, n_hit, n_occl};
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
#ifndef MRT_SMALL_WAVES
#define MRT_SMALL_WAVES 5
#endif
#ifndef MRT_SMALL_TPW_DEFAULT
#define MRT_SMALL_TPW_DEFAULT 1
#endif
// Tile shape (experiment knob, default 8 x 8 lanes).  With two pixels per lane a tile of 16 x 4 lanes writes whole
// 128-byte lines of the ID and t buffers (32 pixels x 4 B) where 8 x 8 lanes write 64-byte halves, and a store-only
// kernel of that shape is faster (tools/ubench_launch.hip, non-temporal: 8.6 against 10.3 us for the 1080p frame) --
// but the flat tiles have more outline for the same area, more of them hold a candidate, and the default frame takes
// 17.05 us instead of 16.25 (make variant NAME=lx16 DEFS=-DMRT_SMALL_LX=16).
#ifndef MRT_SMALL_LX
#define MRT_SMALL_LX 8
#endif
constexpr int small_lanes_across(int P, bool multi) { return (P == 2 && !multi) ? MRT_SMALL_LX : 8; }

template <int P, bool MULTI>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P == 2 && !MULTI ? MRT_SMALL_WAVES : 1, 8)))
void render_small_kernel(uint32_t wh, float cone_ext_s, uint32_t meta, const float *utab, SmallArgs S)
{
  render_small_body<P, MULTI, MRT_SMALL_TPW_DEFAULT, small_lanes_across(P, MULTI)>(wh, cone_ext_s, meta, utab, S);
}

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
    // fields 9..12: the cone-test inputs (the bounce kernel's primary culling, mrt_bounce.hip); 0..3: the screen rectangle below
    prm[9][i] = wx; prm[10][i] = wy; prm[11][i] = wz;
    prm[12][i] = (float)((std::sqrt((double)rr + 2e-6 * L2) * 1.001 + 1e-6) * (1.0 + 1e-6)); // sphere_cull_radius, rounded up
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
  // Spheres, primary rays: the SCREEN RECTANGLE of the tile centres whose (padded) tile can hold a ray that touches
  // the sphere -- four compares per tile in the kernel instead of a cone test.  A ray d hits the ball (w = c - o,
  // radius R) iff (d.w)^2 >= |d|^2 (|w|^2 - R^2); with d(u, v) = base + u h + v vv that is a quadratic S(u, v) >= 0,
  // for a ball in front of the camera the inside of an ellipse, whose extent in u is where max_v S(u, v) = 0 (and
  // likewise in v).  R = the cone test's cull radius plus what the rounding of the ray directions (camera_rounding)
  // can move a ray at the ball's distance.  Anything else -- camera inside the ball, a ball reaching behind the
  // camera (hyperbola), a degenerate camera, NaN -- gets the rectangle that holds every tile.
  {
    double nrm[3] = {h[1] * vv[2] - h[2] * vv[1], h[2] * vv[0] - h[0] * vv[2], h[0] * vv[1] - h[1] * vv[0]};
    const double nl = std::sqrt(dot(nrm, nrm));
    const double plane_dist = nl > 0.0 ? std::fabs(dot(base, nrm)) / nl : 0.0; // every ray is at least this long
    for (int i = 0; i < ns; ++i) {
      float lo_u = -INFINITY, hi_u = INFINITY, lo_v = -INFINITY, hi_v = INFINITY;
      const double w[3] = {prm[9][i], prm[10][i], prm[11][i]};
      const double L = std::sqrt(dot(w, w));
      if (plane_dist > 1e-30 && std::isfinite(L) && std::isfinite((double)prm[12][i])) {
        const double R = ((double)prm[12][i] + L * (cam_delta / plane_dist) * 1.01) * (1.0 + 1e-6) + 1e-6 * L;
        const double K = L * L - R * R;
        auto q = [&](const double *x, const double *y) { return dot(w, x) * dot(w, y) - K * dot(x, y); };
        const double a11 = q(h, h), a12 = q(h, vv), a22 = q(vv, vv), b1 = q(base, h), b2 = q(base, vv), c0 = q(base, base);
        const double det = a11 * a22 - a12 * a12;
        if (K > 0.0 && a11 < 0.0 && a22 < 0.0 && det > 1e-9 * a11 * a22) {
          const double uc = (a12 * b2 - a22 * b1) / det, vc = (a12 * b1 - a11 * b2) / det; // the ellipse's centre
          const double dc[3] = {base[0] + uc * h[0] + vc * vv[0], base[1] + uc * h[1] + vc * vv[1], base[2] + uc * h[2] + vc * vv[2]};
          const double Au = a11 - a12 * a12 / a22, Bu = b1 - a12 * b2 / a22, Cu = c0 - b2 * b2 / a22;
          const double Av = a22 - a12 * a12 / a11, Bv = b2 - a12 * b1 / a11, Cv = c0 - b1 * b1 / a11;
          const double du = Bu * Bu - Au * Cu, dv = Bv * Bv - Av * Cv;
          if (dot(w, dc) > 0.0 && Au < 0.0 && Av < 0.0 && du > 0.0 && dv > 0.0) { // in front; real extents
            const double su = std::sqrt(du) / -Au, sv = std::sqrt(dv) / -Av, mu = -Bu / Au, mv = -Bv / Av;
            const double pad_u = hu + 1e-5 * (1.0 + std::fabs(mu) + su), pad_v = hv + 1e-5 * (1.0 + std::fabs(mv) + sv);
            lo_u = std::nextafter((float)(mu - su - pad_u), -INFINITY); hi_u = std::nextafter((float)(mu + su + pad_u), INFINITY);
            lo_v = std::nextafter((float)(mv - sv - pad_v), -INFINITY); hi_v = std::nextafter((float)(mv + sv + pad_v), INFINITY);
            if (!(lo_u <= hi_u) || !(lo_v <= hi_v)) { lo_u = lo_v = -INFINITY; hi_u = hi_v = INFINITY; }
          }
        }
      }
      prm[0][i] = lo_u; prm[1][i] = hi_u; prm[2][i] = lo_v; prm[3][i] = hi_v;
    }
  }
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

// Camera-dependent kernel arguments of a small scene for tiles of lx * P x 64 / lx pixels: cone extent and prm.
void small_prepare_args(const RenderArgs &in, const SmallHost &hs, int P, SmallArgs &S, int lx)
{
  S.A = in;
  RenderArgs &A = S.A;
  // the padded tile (pixels -1..lx P horizontally, -1..64 / lx vertically) around its centre, in u / v units
  const double hw = (0.5 * lx * P + 0.5) / A.W, hh = (0.5 * (64 / lx) + 0.5) / A.H;
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

// Tiles per wave: horizontally adjacent tiles can share one prologue (compile-time MRT_SMALL_TPW_DEFAULT).  Measured
// on the default frame: more than one makes the frame SLOWER (2: +18 %, 4: +59 %) -- the waves of candidate tiles
// live twice as long and the frame ends when the last of them does.
static int small_tiles_per_wave() { return MRT_SMALL_TPW_DEFAULT; }

// The tile row the launch starts with (the kernel wraps around).  Decided from the view itself, on the host.  This is
// the rule as it was first built and measured, with the kernel's own per-tile tests (S.prm) -- 16 200 tiles x 10
// primitives for the default frame, a few hundred microseconds, which only a view that is rendered AGAIN could afford;
// round 3 put small_first_row_fast (below) in its place for every new view and keeps this one as its yardstick
// (MRT_SMALL_ROW0_DEBUG).  Rule: skip the leading rows that
// hold less than 2 % of the frame's candidate tiles, keep a margin of 1/20 of the rows before the geometry begins
// (measured on the default frame, whose geometry begins in row 28 of 135: start rows 20-26 15.7-15.9 us, 0: 16.5,
// 28: 16.4, 30: 16.9 -- the rows skipped run LAST, and a candidate tile that starts last is a tail).  Results do
// not depend on the order.  MRT_SMALL_ROW0 = n pins the row (experiments), -1 = this rule.
static int small_first_row(const SmallArgs &S, int ns, int nt, int P, int lx, int tiles_x, int tiles_y)
{
  const int TW = lx * P, LY = 64 / lx;
  const float rw = 1.0f / float(S.A.W), rh = 1.0f / float(S.A.H);
  // a bounded amount of host work whatever the frame: every xs-th column of tiles (the rule below only compares counts)
  const long work = (long)tiles_x * tiles_y * (ns + nt > 0 ? ns + nt : 1);
  const int xs = int(work / 2000000L) + 1;
  // candidate tiles of one row.  No heap here: this runs inside mrt_render_device_static, which promises to call no
  // allocating routine (include/mrt.h) -- the rows are counted twice (total, then cumulative) instead of being kept.
  auto row_candidates = [&](int y) {
    const float vm = (float(y * LY) + 0.5f * float(LY - 1)) * rh;
    int c = 0;
    for (int x = 0; x < tiles_x; x += xs) {
      const float um = (float(x * TW) + 0.5f * float(TW - 1)) * rw;
      bool any = false;
      for (int i = 0; i < ns && !any; ++i)
        any = !(um < S.prm[0][i]) & !(um > S.prm[1][i]) & !(vm < S.prm[2][i]) & !(vm > S.prm[3][i]);
      for (int j = ns; j < ns + nt && !any; ++j) {
        bool cull = false;
        for (int i = 0; i < 4; ++i)
          cull |= std::fmaf(um, S.prm[3 * i + 1][j], std::fmaf(vm, S.prm[3 * i + 2][j], S.prm[3 * i][j])) < 0.0f;
        any = !cull;
      }
      c += any ? 1 : 0;
    }
    return c;
  };
  long total = 0;
  for (int y = 0; y < tiles_y; ++y) total += row_candidates(y);
  if (total == 0) return 0;
  long cum = 0;
  int first = 0;
  for (; first < tiles_y; ++first) {
    cum += row_candidates(first);
    if (cum * 50 > total) break; // more than 2 % of the candidate tiles lie in rows 0..first
  }
  const int margin = tiles_y / 20 > 2 ? tiles_y / 20 : 2;
  return first > margin ? first - margin : 0;
}

// The same rule from an ESTIMATE of the rows' candidate counts that costs microseconds instead of hundreds of them, so
// that the first render of a view -- and every render of a camera that moves each frame -- gets the tuned start row
// too.  Per sampled row of tiles and primitive, the columns whose tile centre passes the primitive's test are an
// interval: for a sphere the u side of its screen rectangle, for a triangle the intersection of the four half-planes
// W0' + um Wh + vm Wv >= 0 at the row's vm (one division each).  The row's count is the sum of the intervals' lengths
// (tiles under several primitives count several times: the rule only compares counts).  At most 256 rows are sampled;
// no heap (this also runs inside mrt_render_device_static).
static int small_first_row_fast(const SmallArgs &S, int ns, int nt, int P, int lx, int tiles_x, int tiles_y)
{
  const int TW = lx * P, LY = 64 / lx;
  const float rw = 1.0f / float(S.A.W), rh = 1.0f / float(S.A.H);
  const float um0 = 0.5f * float(TW - 1) * rw, dum = float(TW) * rw, inv_dum = 1.0f / dum; // um(x) = um0 + x dum
  const int ys = (tiles_y + 255) / 256;
  float cnt[256];
  int nrow = 0;
  auto columns = [&](float ulo, float uhi) { // tile columns whose centre lies in [ulo, uhi]
    if (!(ulo <= uhi)) return 0.0f;          // (empty, or NaN)
    float xlo = std::ceil((ulo - um0) * inv_dum), xhi = std::floor((uhi - um0) * inv_dum);
    xlo = xlo > 0.0f ? xlo : 0.0f;
    xhi = xhi < float(tiles_x - 1) ? xhi : float(tiles_x - 1);
    return xhi >= xlo ? xhi - xlo + 1.0f : 0.0f;
  };
  double total = 0.0;
  for (int y = 0; y < tiles_y && nrow < 256; y += ys) {
    const float vm = (float(y * LY) + 0.5f * float(LY - 1)) * rh;
    float c = 0.0f;
    for (int i = 0; i < ns; ++i)
      if (!(vm < S.prm[2][i]) && !(vm > S.prm[3][i])) c += columns(S.prm[0][i], S.prm[1][i]);
    for (int j = ns; j < ns + nt; ++j) {
      float ulo = -INFINITY, uhi = INFINITY;
      for (int i = 0; i < 4; ++i) {
        const float a = S.prm[3 * i + 1][j], b = std::fmaf(vm, S.prm[3 * i + 2][j], S.prm[3 * i][j]);
        if (a > 0.0f) ulo = std::fmax(ulo, -b / a);
        else if (a < 0.0f) uhi = std::fmin(uhi, -b / a);
        else if (b < 0.0f) uhi = -INFINITY; // the function does not depend on u and culls the whole row (NaN: never culls)
      }
      c += columns(ulo, uhi);
    }
    cnt[nrow++] = c;
    total += c;
  }
  if (!(total > 0.0)) return 0;
  double cum = 0.0;
  int first = 0;
  for (; first < nrow; ++first) {
    cum += cnt[first];
    if (cum * 100.0 > total) break; // more than 1 % of the (estimated) candidate tiles lie in the sampled rows up to this one
  }
  first *= ys;
  // (tiles under several primitives weigh more here than in the per-tile count, which moves the threshold row down the
  // frame by a few rows on the default view: a lower threshold and a wider margin than the per-tile rule's 2 % and 1/20)
  const int margin = tiles_y / 16 > 2 ? tiles_y / 16 : 2;
  return first > margin ? first - margin : 0;
}

static int small_row0(const SmallArgs &S, const SmallHost &hs, int P, int lx, int tiles_x, int tiles_y, bool partitioned)
{
  static const int forced = [] { const char *e = getenv("MRT_SMALL_ROW0"); return e ? atoi(e) : -1; }();
  if (forced >= 0) return forced % tiles_y;
  SmallViewCache *vc = hs.view;
  if (partitioned || tiles_y < 16) return 0; // strips of a partitioned frame are dealt over the ranks as they are
  if (!vc) return small_first_row_fast(S, hs.ns, hs.nt, P, lx, tiles_x, tiles_y); // no memory of views: every render is a first one
  SmallViewCache::Key k;
  std::memset(&k, 0, sizeof k);
  k.cam = S.A.cam; k.W = S.A.W; k.H = S.A.H; k.local_rows = S.A.local_rows; k.P = P; k.lx = lx; k.partitioned = 0;
  static const int dbg = getenv("MRT_SMALL_ROW0_DEBUG") ? atoi(getenv("MRT_SMALL_ROW0_DEBUG")) : 0; // 1: print both rules' rows; 2: use the exact rule
  if (vc->valid && std::memcmp(&k, &vc->key, sizeof k) == 0) {
    if (vc->seen < (1 << 30)) ++vc->seen;
    if (dbg == 2 && vc->seen == 2) vc->row0 = small_first_row(S, hs.ns, hs.nt, P, lx, tiles_x, tiles_y);
    return vc->row0 < tiles_y ? vc->row0 : 0;
  }
  // a view not seen before: the estimate (microseconds); rendered again, the view keeps its row
  vc->key = k; vc->valid = true; vc->seen = 1;
  vc->row0 = small_first_row_fast(S, hs.ns, hs.nt, P, lx, tiles_x, tiles_y);
  if (dbg == 1)
    fprintf(stderr, "[mrt] start row: estimate %d, per-tile rule %d (of %d rows)\n", vc->row0,
            small_first_row(S, hs.ns, hs.nt, P, lx, tiles_x, tiles_y), tiles_y);
  return vc->row0 < tiles_y ? vc->row0 : 0;
}

template <int P>
static hipError_t launch_small(const RenderArgs &in, const SmallHost &hs, hipStream_t stream)
{
  SmallArgs S; // 3.5 KB of kernel arguments
  const int lx = small_lanes_across(P, in.spp > 1), ly = 64 / lx;
  small_prepare_args(in, hs, P, S, lx);
  RenderArgs &A = S.A;
  // vector stores need every lane's first pixel on a 4P-byte boundary in all three buffers
  const uintptr_t align = 4 * P - 1;
  A.wide = (A.W % P == 0) && ((uintptr_t)A.rgb & align) == 0 && ((uintptr_t)A.id & align) == 0 &&
           ((uintptr_t)A.t & align) == 0;
  const int tiles_x = (A.W + lx * P - 1) / (lx * P), tiles_y = (A.local_rows + ly - 1) / ly;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  // the arguments the hardware preloads (SmallHot): camera in culling form, frame size, cone extent (its sign flags a
  // strip partition), the u table, tiles per wave and primitive counts
  const int tpw = small_tiles_per_wave();
  const bool partitioned = !(A.nranks == 1 && A.rank == 0);
  const float cone_s = partitioned ? -A.cone_ext : A.cone_ext;
  const uint32_t wh = (uint32_t)(A.W - 1) | ((uint32_t)(A.H - 1) << 16);
  constexpr int BANDS = 1 << MRT_SMALL_BANDS_LOG2;
  int band_rows = (tiles_y + BANDS - 1) / BANDS; // < 65536: tiles_y <= 8192
  const int grid_rows = band_rows * BANDS;
  if (BANDS == 1) band_rows = small_row0(S, hs, P, lx, tiles_x, tiles_y, partitioned);
  const uint32_t meta = (uint32_t)A.sc.ns | ((uint32_t)A.sc.nt << 8) | ((uint32_t)band_rows << 16);
  // one wave per workgroup: waves share nothing, and single-wave workgroups fill freed wave slots sooner than
  // groups of four (measured 19.4 against 19.9 us on the default frame)
  const dim3 grid((tiles_x + tpw - 1) / tpw, grid_rows);
  if (A.spp > 1)
    hipLaunchKernelGGL((render_small_kernel<P, true>), grid, dim3(64), 0, stream, wh, cone_s, meta, A.utab, S);
  else
    hipLaunchKernelGGL((render_small_kernel<P, false>), grid, dim3(64), 0, stream, wh, cone_s, meta, A.utab, S);
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

#ifdef MRT_SMALL_ORDER_TABLE
extern "C" int mrt_diag_set_order(const unsigned short *rows, size_t n)
{
  if (n > 8192) return -1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_order), rows, n * sizeof(unsigned short), 0, hipMemcpyHostToDevice);
}
#endif
#ifdef MRT_TIMELINE
extern "C" int mrt_diag_timeline(unsigned long long *host_out, size_t words)
{
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

hipError_t launch_render_small(const RenderArgs &A, const SmallHost &hs, hipStream_t stream)
{
  // the kernel finds the v table behind the u table (uv_table_stride) and packs the frame size into 2 x 16 bits
  if (A.sc.ns + A.sc.nt > 64 || !A.utab || A.vtab != A.utab + uv_table_stride(A.W) || hs.ns != A.sc.ns ||
      hs.nt != A.sc.nt || A.W > 65536 || A.H > 65536)
    return hipErrorInvalidValue;
  switch (small_pixels_per_lane(A.spp)) {
  case 1: return launch_small<1>(A, hs, stream);
  case 4: return launch_small<4>(A, hs, stream);
  default: return launch_small<2>(A, hs, stream);
  }
}

} // namespace mrt
