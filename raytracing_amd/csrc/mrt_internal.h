// mrt_internal.h -- shared between the C-ABI (mrt_api.cpp) and the gfx950 kernels (mrt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

namespace mrt {

// Scene as it lives in HBM: pure SoA streams, one allocation.  Primitives of a type are stored in kd
// order of their bound centres (mrt_scene_build.h); *_obj carries the object number every result is reported in.
struct DevScene {
  // spheres: 4 streams of ns floats (core/sphere.cuh:116-117) + object numbers
  const float *sph_cx, *sph_cy, *sph_cz, *sph_r;
  const int32_t *sph_obj;
  int32_t ns;
  // triangles: 16 streams of nt floats, stream s at tri + s*nt (core/triangle.cuh:118-122)
  const float *tri;
  const int32_t *tri_obj;
  // conservative bounding spheres of the regions the triangle hit test accepts (built at upload,
  // mrt_scene_build.h tri_bound); radius +inf = never culled.  4 streams of nt floats.
  const float *tri_bx, *tri_by, *tri_bz, *tri_br;
  // the same 16 floats per triangle, contiguous (64 B records), for wave-uniform fetches of the few
  // triangles that survive culling (one s_load_dwordx16 instead of 16 strided scalar loads)
  const float *tri_rec;
  // shadow-ray edge functions of the first nls lights, [light][23][triangle] (scenes of <= 64 primitives only;
  // mrt_scene_build.h shadow_functions, used by mrt_small.hip)
  const float *tri_shfn;
  int32_t nls;
  // bounding balls of runs of 64 consecutive primitives (primitives are stored in kd order):
  // 4 streams (x, y, z, r) of ncs / nct floats each
  const float *sph_clu, *tri_clu;
  int32_t ncs, nct;
  // ... and of runs of 64 clusters ("super-clusters")
  const float *sph_sclu, *tri_sclu;
  int32_t nscs, nsct;
  int32_t nt;
  int32_t nct_regular; // triangle clusters ahead of the first one that holds loose / unbounded triangles only (== nct: none)
  // per object material (core/material.cuh:78,160)
  const uint8_t *mat_kind;
  const float *mat_rgb; // [nobj*3]
  int32_t nobj;
  // lights (core/light.cuh:56-57, :84-85)
  const uint8_t *light_kind;
  const float *light_vec; // [nl*3]
  const float *light_rgb; // [nl*3]
  int32_t nl;
};

struct DevCamera {
  float ox, oy, oz;
  float hx, hy, hz;
  float vx, vy, vz;
  float ux, uy, uz; // up_left
};

// Bound on the absolute fp32 rounding error of a primary ray direction, ((ul + u h) + v vv) - o with u, v in
// about [0, 1]: five roundings, each relative to at most the sum of the magnitudes involved.  Negligible next to a
// pixel while the camera sits within a few thousand units of the world origin; far out (|o| ~ 1e5) the computed
// directions are coarser than the pixel grid and the culling cones and edge functions have to allow for it.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline float camera_rounding(const DevCamera &c)
{
  const float mag = fabsf(c.ux) + fabsf(c.uy) + fabsf(c.uz) + fabsf(c.hx) + fabsf(c.hy) + fabsf(c.hz) + fabsf(c.vx) +
                    fabsf(c.vy) + fabsf(c.vz) + fabsf(c.ox) + fabsf(c.oy) + fabsf(c.oz);
  return 9.6e-7f * mag; // 16 x 2^-24
}

struct RenderArgs {
  DevScene sc;
  DevCamera cam;
  int32_t W, H;
  int32_t rank, nranks, strip_rows; // strip partition; rank: bits 0-15 the rank, bits 16-31 its span - 1 (global_strip below)
  int32_t local_rows;
  int32_t frame_layout;             // 0: buffers hold the rank's rows back to back; 1: whole frames, rows at their final place
  int32_t spp;                      // 1 or 121
  int32_t max_depth;                // 0: primary + shadow; 1..8: mirror chains (mrt_bounce.hip)
  float *rgb;                       // [local_rows*W*3]
  int32_t *id;                      // [local_rows*W] or null
  float *t;                         // [local_rows*W] or null
  unsigned long long *stats;        // stats_slots x 8 counters (the host adds the slots up) or null
  int32_t stats_slots;              // power of two; 1 for a caller's 8-counter buffer
  // LDS plan (filled by the launcher)
  int32_t resident;                 // whole scene staged once
  int32_t s_cap, t_cap;             // primitives per LDS chunk
  // small-scene kernel (mrt_small.hip): u = col / W and v = row / H tables, vector-store switch
  const float *utab, *vtab;
  int32_t wide;
  float cone_ext;                   // cull kernel: max |ray - tile centre ray| over a padded tile (launcher)
  int32_t tile_map;                 // cull kernel: 0 tiles in row order (2-D grid), n: XCD-aware blocks of 2^n x 2^n tiles (1-D grid)
  const uint32_t *block_order;      // with tile_map: the blocks in the order they are to start (heaviest first), or null: as numbered
  int32_t team_mode;                // launcher options (opt_team / opt_lists / opt_list_cap below); MRT_OPT_CULL_TEAM: 0 automatic, 1 one wave per tile, 4 teams dealing leaves, 5 dealing lights
  // diffuse bounces (mrt_bounce.hip, mrt_rng.hip): xorshift words of pixel i at rng[k * rng_stride + i], k = 0..4,
  // and the Weyl counter every stream starts with; null when no stream is needed
  const uint32_t *rng;
  uint32_t rng_stride, rng_d;
  // bounce kernel: wave w renders tile tile_order[w] (null: tile w); tiles are numbered row by row
  const uint32_t *tile_order;
  // cull kernel, primary rays: the loose triangles (those from cluster loose_nct_walk on) per block of 64 x 64 pixels of
  // the frame -- loose_cnt[block] entries (16-bit offsets from the first loose triangle) at loose_list + block *
  // loose_cap; a count above loose_cap = the list overflowed, take them all.  Null: no lists, the walk covers everything.
  const uint16_t *loose_list;
  const uint32_t *loose_cnt;
  int32_t loose_bx, loose_cap;          // blocks per row of blocks; entries per block
  int32_t loose_nct_walk, loose_nsct_walk; // clusters / super-clusters the hierarchy walk still covers
  // cull kernel, primary rays: per-tile lists of the primitives whose bound the tile's cone may touch (mrt_cull.hip,
  // bin_primitives_kernel).  bin_cnt: BIN_HEAD control words, then two counters per tile of the whole frame (spheres,
  // triangles); bin_list: bin_cap entries per counter; bin_near: primitives around the camera, for every tile.
  // Null: no bins, every tile walks the hierarchy.
  const uint32_t *bin_cnt;
  const uint32_t *bin_list;
  const uint32_t *bin_near;
  int32_t bin_cap;
  int32_t bin_nct_regular;              // triangle clusters the bins cover (the loose ones behind them are not binned)
  int32_t bin_clusters;                 // 0: the lists hold primitives; 1: leaf clusters of 64 (scenes of more primitives than tiles)
  int32_t solo_max;                     // teams dealing leaves: a tile with at most this many batches listed is rendered by member 0 alone
};
// A rank may render the strips of SPAN consecutive ranks of the partition (mrt_partition.rank_span: rank 0 of a weighted
// multi-GPU frame takes several shares).  Its local strips, in ascending order, are the frame's strips s with
// rank <= s % nranks < rank + span; RenderArgs.rank carries the span in its upper half.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int pack_rank(int rank, int span) { return rank | ((span > 1 ? span - 1 : 0) << 16); }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int global_strip(int local_strip, int packed_rank, int nranks)
{
  const int span = (packed_rank >> 16) + 1, rank = packed_rank & 0xffff;
  if (span == 1) return local_strip * nranks + rank;
  return (local_strip / span) * nranks + rank + local_strip % span;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline bool owns_strip(int strip, int packed_rank, int nranks)
{
  const int span = (packed_rank >> 16) + 1, rank = packed_rank & 0xffff, r = strip % nranks;
  return r >= rank && r < rank + span;
}

// (SmallArgs below is at the 4 KB limit of a kernel's arguments, so the launcher-only options share team_mode's word:
// bits 0-7 MRT_OPT_CULL_TEAM; bits 8-9 MRT_OPT_LISTS, 0 by the scene, 1 no lists at all (every tile walks), 2 lists of
// primitives, 3 lists of leaf clusters; bits 16-30 MRT_OPT_LIST_CAP, > 0 (test hook) shortens every list to reach the
// overflow paths)
inline int opt_team(const RenderArgs &A) { return A.team_mode & 0xff; }
inline int opt_lists(const RenderArgs &A) { return (A.team_mode >> 8) & 3; }
inline int opt_list_cap(const RenderArgs &A) { return (A.team_mode >> 16) & 0x7fff; }

hipError_t launch_render_cull(const RenderArgs &args, hipStream_t stream);
// per-frame scratch of the loose triangles' block lists (device; loose_scratch_bytes() of it, or null = no lists)
size_t loose_scratch_bytes(const RenderArgs &args);
hipError_t launch_render_cull(const RenderArgs &args, void *loose_scratch, hipStream_t stream);

// scenes of at most 64 primitives (mrt_small.hip); needs args.utab / args.vtab
// host copy of a small scene in device order, from which the launcher derives the camera-dependent kernel arguments
// What the small-scene launcher remembers about the last view it rendered (owned by the context, reset by an
// upload): a caller that renders the same view again -- a benchmark loop, progressive accumulation, a paused viewer --
// gets the launch order of its tile rows tuned to where that view's geometry is (mrt_small.hip, small_first_row).
struct SmallViewCache {
  struct Key { DevCamera cam; int W, H, local_rows, P, lx, partitioned; } key;
  bool valid = false;
  int seen = 0;   // renders of this view so far
  int row0 = 0;   // first tile row to launch (small_first_row_fast, worked out when the view is first seen)
};
struct SmallHost {
  SmallViewCache *view = nullptr; // optional
  int ns = 0, nt = 0;
  const float *sph = nullptr;       // [ns][4] centre xyz, radius
  const int32_t *sph_obj = nullptr; // [ns]
  const float *tri16 = nullptr;     // [nt][16]
  const int32_t *tri_obj = nullptr; // [nt]
};
hipError_t launch_render_small(const RenderArgs &args, const SmallHost &host, hipStream_t stream);

// Kernel arguments of the small-scene kernels: the common ones plus everything a wave needs about "its"
// primitives (lane = primitive: spheres first, then triangles) before the first exact test, [field][lane]:
//   sphere lanes:   0-3 screen rectangle (lo u, hi u, lo v, hi v) of the tile centres whose padded tile can hold a ray
//                   that touches the sphere; 4-6 origin - centre, 7 |o - c|^2 - r^2, 8 object; 9-11 centre - origin and
//                   12 primary cull radius (the cone test of the bounce kernels' primary iterations)
//   triangle lanes: 0-11 four functions (3 edges + plane side) x (W0', Wh, Wv): "W0' + um Wh + vm Wv < 0" at the
//                   tile centre (um, vm) means that no ray of the padded tile passes that test;
//                   12 k - n.o, 13 object
// The exact-test terms (4-7, 12) are computed on the host with the kernels' fp32 operations in the same order.
enum { SMALL_FIELDS = 14 };
struct SmallArgs {
  RenderArgs A;
  float prm[SMALL_FIELDS][64];
};
static_assert(sizeof(SmallArgs) <= 4096, "kernel arguments are limited to 4 KB");
// fills S for tiles of 8P x 8 pixels (mrt_small.hip): S.A = args with cone_ext set, S.prm
void small_prepare_args(const RenderArgs &args, const SmallHost &host, int P, SmallArgs &S, int lanes_across = 8);
hipError_t launch_uv_table(float *utab, int nu, int W, float *vtab, int nv, int H, hipStream_t stream);
// floats between the start of the u table and the start of the v table behind it: whole tiles of up to 32 pixels,
// plus room for the one-tile-ahead fetches of a wave's last tile (up to 4 tiles of 32 pixels per wave)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int uv_table_stride(int W) { return ((W + 31) & ~31) + 128; }

// bounce chains, max_depth 1..8 (mrt_bounce.hip); a scene with a diffuse object needs args.rng; host != null
// (scenes of at most 64 primitives) lets the primary rays use the small-scene kernel's per-tile culling
// scratch (device, bounce_scratch_words() words, or null) lets the launcher order the tiles by predicted cost
hipError_t launch_render_bounce(const RenderArgs &args, const SmallHost *host, uint32_t *scratch, hipStream_t stream);
hipError_t launch_render_bounce_pipe(const SmallArgs &S, hipStream_t stream); // mrt_bounce_pipe.hip
enum { BOUNCE_SCRATCH_HEAD = 128 }; // words ahead of the per-tile arrays: bucket counters and cursors
inline size_t bounce_scratch_words(int W, int local_rows) { return 4 * (size_t)((W + 7) / 8) * ((local_rows + 7) / 8) + BOUNCE_SCRATCH_HEAD; }

// per-pixel XORWOW streams (mrt_rng.hip): jump matrices A^(2^(67+b)), b < RNG_JUMP_MATRICES, 800 words each
enum { RNG_JUMP_MATRICES = 32 };
const uint32_t *rng_jump_tables();
void rng_seed_state(uint64_t seed, uint32_t v[5], uint32_t *d);
void rng_host_state(uint64_t seed, uint32_t subsequence, uint32_t out6[6]);
hipError_t launch_rng_init(const uint32_t *jump_dev, uint32_t *states, uint32_t n, uint32_t stride, uint64_t seed,
                           hipStream_t stream);

// launchers (mrt_kernels.hip)
hipError_t launch_render(const RenderArgs &args, hipStream_t stream);
// mode: bit 0 the interval's lower bound is open, bit 1 its upper bound (core/interval.cuh:14-23)
hipError_t launch_intersect(int prim_kind, const float *prim_dev, const float *rays_dev, float tmin, float tmax, int mode,
                            int64_t n, uint8_t *hit_dev, float *t_dev, float *point_dev, float *normal_dev,
                            hipStream_t stream);
hipError_t launch_stats_reduce(const unsigned long long *slots, int nslots, unsigned long long *out8,
                               hipStream_t stream);
// rank 0 after the gather (mrt_scatter_strips): packed per-rank buffers -> the three frame planes
enum { MAX_RANKS = 64 };
struct ScatterArgs {
  const uint32_t *staging;
  uint32_t *rgb, *id, *t;          // frame planes (id / t may be null)
  int32_t W, H, nranks, strip_rows, skip_rank, skip_span; // ranks skip_rank .. skip_rank + skip_span - 1 rendered in place
  int32_t has_id, has_t;           // which planes follow rgb in a packed buffer (MRT_PLANE_ID / MRT_PLANE_T)
  int32_t rank_rows[MAX_RANKS];    // rows of rank r
  uint64_t rank_off[MAX_RANKS];    // dword offset of rank r's packed buffer in staging
};
hipError_t launch_scatter(const ScatterArgs &S, hipStream_t stream);

} // namespace mrt
