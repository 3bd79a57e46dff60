/*
 * mrt.h -- C-ABI of libmrt.so, the MI355X (gfx950) back end of the render hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI; its de-facto boundary
 * is the set of CUDA-runtime touch points funnelled through include/cuda_utils/
 * and src/check_error.cu (reference paths below are relative to its checkout).
 * Each entry point cites the touch point it replaces.  HIP is visible only
 * inside the library: signatures carry plain pointers and sizes.
 *
 * Conventions
 *  - every function returns MRT_OK (0) or a negative mrt_status; nothing exits
 *    or throws (the fail-fast print-and-exit convention of src/check_error.cu:7-15
 *    lives on the C++ side as checkRtErrors(), raytracing_amd/host/hip_utils/check_error.hpp);
 *  - one context drives ONE GPU; multi-GPU frames are split in strips (mrt_partition)
 *    and gathered on rank 0 over RCCL (mrt_comm_*, mrt_gather_strips, mrt_render_multi),
 *    one process per GPU or one process over several contexts;
 *  - calls on one context are serialised by the caller (the reference is single
 *    threaded, src/main.cu:159-260).  Renders of one context share its counters and
 *    scratch, so the library orders them on the GPU: a render enqueued on another
 *    stream than the previous one first waits (on the GPU, not the host) for that
 *    one.  Renders of DIFFERENT contexts overlap freely.  mrt_scene_upload and
 *    mrt_ctx_destroy wait for the context's own stream and for the stream of its
 *    last render; work the caller enqueued elsewhere is the caller's to synchronise;
 *  - all arithmetic is fp32 with FMA contraction off and IEEE div/sqrt, in the
 *    operation order documented in DESIGN.md, so hit/ID buffers are bit-comparable
 *    with the CPU oracle.
 */
#ifndef MRT_H
#define MRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum mrt_status {
  MRT_OK = 0,
  MRT_ERR_INVALID = -1,   /* bad argument (null pointer, size mismatch, unsupported mode) */
  MRT_ERR_HIP = -2,       /* a HIP runtime call failed; mrt_last_error() has the text */
  MRT_ERR_NO_SCENE = -3,  /* render before mrt_scene_upload */
  MRT_ERR_NO_DEVICE = -4, /* no usable GPU */
  MRT_ERR_TOO_LARGE = -5  /* size beyond a documented limit */
} mrt_status;

enum { MRT_GEOM_SPHERE = 0, MRT_GEOM_PLANE = 1, MRT_GEOM_TRIANGLE = 2 }; /* core/scene_object.cuh:14 */
enum { MRT_MAT_DIFFUSE = 0, MRT_MAT_METALLIC = 1 };                      /* core/scene_object.cuh:16 */
enum { MRT_LIGHT_DIRECTIONAL = 0, MRT_LIGHT_POINT = 1 };                 /* core/light.cuh:60, :36 */

typedef struct mrt_ctx mrt_ctx;

/*
 * Scene as the reference's kernel consumes it, flattened to SoA host arrays.
 * Replaces rt::Vector<rt::SceneObject> / rt::Vector<rt::DirectionalLight> in
 * managed memory (container/vector.cuh:15-177, built at src/main.cu:210-237).
 * The library copies everything; the caller keeps ownership.
 *
 * Objects are numbered 0..num_objects-1 in scene order; closest-hit ties and
 * the ID buffer are defined on that number (src/main.cu:67-77).
 * Planes (core/plane.cuh:25-29) never hit, so they only occupy an object number.
 */
typedef struct mrt_scene_desc {
  int32_t num_objects;
  const uint8_t *mat_kind;  /* [num_objects] MRT_MAT_* */
  const float *mat_rgb;     /* [num_objects*3] material colour (core/material.cuh:78, :160) */

  int32_t num_spheres;
  const float *sphere_cx, *sphere_cy, *sphere_cz, *sphere_r; /* [num_spheres] each (core/sphere.cuh:116-117) */
  const int32_t *sphere_obj;                                 /* [num_spheres] object number, ascending */

  int32_t num_triangles;
  /* 16 streams of [num_triangles], stream s at tri + s*num_triangles: the state the
   * Triangle constructor leaves behind (core/triangle.cuh:58-66, :118-122):
   * 0-2 plane normal, 3 plane intercept, then three edge discriminators
   * (normal xyz, intercept) for ab|c, bc|a, ca|b */
  const float *tri;
  const int32_t *tri_obj; /* [num_triangles] object number, ascending */

  int32_t num_lights;
  const uint8_t *light_kind; /* [num_lights] MRT_LIGHT_* */
  const float *light_vec;    /* [num_lights*3] unit direction (directional) or position (point) */
  const float *light_rgb;    /* [num_lights*3] */
} mrt_scene_desc;

/* rt::Viewport after construction (pipeline/viewport.cuh:43-47), computed on the host */
typedef struct mrt_camera {
  float origin[3];
  float horizontal[3];
  float vertical[3];
  float up_left[3];
} mrt_camera;

/*
 * Strip partition of a frame over ranks (no reference counterpart: the reference is
 * single GPU).  Rows are dealt in strips of strip_rows rows,
 * strip s -> rank s % num_ranks.  A rank's local buffer holds its strips back to
 * back in ascending order (frame_layout 0).  {0,1,8,0,0} renders the whole frame.
 */
typedef struct mrt_partition {
  int32_t rank;
  int32_t num_ranks;
  int32_t strip_rows;   /* multiple of 8 */
  int32_t frame_layout; /* 0: the buffers hold the rank's rows back to back (local_rows x W);
                           1: the buffers are whole H x W frames and the rank writes only its own rows, each at its
                              final place (rank 0 of a multi-GPU frame renders like this and receives the rest) */
  int32_t rank_span;    /* 0 or 1: the strips of `rank` alone.  k > 1: the strips of ranks rank .. rank + k - 1 in ONE
                           render (in ascending strip order) -- how rank 0 of a WEIGHTED multi-GPU frame takes several
                           shares of a partition into more shares than GPUs (mrt_gather_strips_root).  At most 65535
                           ranks; rank + k <= num_ranks */
} mrt_partition;

typedef struct mrt_render_params {
  int32_t width, height;  /* full frame */
  int32_t spp;            /* 1: one sample at (col/W,row/H); 121: rt::MultisampleMesh (pipeline/multisample_mesh.cuh) */
  int32_t max_depth;      /* config::MAX_RECURSION_DEPTH (src/main.cu:26; the reference is built with 8);
                             0 = primary + shadow; 1..8 = bounce chains (src/main.cu:79-98): metallic surfaces
                             mirror, diffuse ones draw the bounce from the pixel's XORWOW stream
                             (core/material.cuh:52-61, cuda_utils/curand_state.cuh:15-23; MRT_OPT_RNG_SEED) */
  mrt_partition part;
} mrt_render_params;

typedef struct mrt_stats {
  uint64_t primary_rays; /* local pixels * spp */
  uint64_t shadow_rays;  /* lights passing the L.N >= 0 test (core/light.cuh:148) */
  uint64_t hit_pixels;
  uint64_t occluded;
  uint64_t exact_tests;     /* wave-level exact primitive tests executed (culling kernels; diagnostic) */
  uint64_t clusters_opened; /* 64-primitive clusters whose members were bound-tested (one-level kernel) */
  float kernel_ms;       /* HIP-event time of the render kernel(s) */
} mrt_stats;

/* --- device / context: replaces the implicit device 0 + cudaDeviceReset of src/check_error.cu:12 */
int mrt_device_count(int *count);
int mrt_ctx_create(int device, mrt_ctx **out);
int mrt_ctx_destroy(mrt_ctx *ctx);
/* last error text of this context (or of the failed create when ctx is NULL); never NULL */
const char *mrt_last_error(const mrt_ctx *ctx);
const char *mrt_status_string(int status);

/*
 * Context options.  MRT_OPT_CULL (no reference counterpart; default 1, or environment MRT_CULL) picks the
 * render kernel: 0 = brute force, which like the reference loops (src/main.cu:67-77) tests every ray against
 * every primitive; 1 = automatic; 2 = tile culling over the primitive hierarchy; 5 = the small-scene kernel (at
 * most 64 primitives, else 2).  All kernels produce identical buffers.  MRT_OPT_RNG_SEED (default 1984, the
 * constant of cuda_utils/curand_state.cuh:22; 0..2^31-1) seeds the per-pixel random streams of diffuse bounces:
 * pixel i of the full frame owns subsequence i, like curand_init(seed, i, 0, ...).
 * MRT_OPT_CULL_TEAM (no reference counterpart; default 0, or environment MRT_CULL_TEAM): waves per tile in the tile-culling
 * kernel -- 0 = by the scene, 1 = one wave per tile, 4 = four waves share a tile's leaf clusters, 5 = four waves share a
 * tile's lights (scenes with many lights).  Identical buffers in every mode.
 * MRT_OPT_FAIL_NEXT_UPLOAD (test hook; value 1) makes the next mrt_scene_upload of the context fail at its
 * device allocation, after the old scene has been given up: the context must then report MRT_ERR_NO_SCENE.
 * MRT_OPT_VIEW_CACHE (no reference counterpart; default 1): the small-scene kernel's launcher starts the launch at the
 * tile row where the view's geometry begins (a scheduling choice: identical buffers), which it works out on the host
 * for every new view (camera + frame size; a few microseconds) and remembers while the same view is rendered again.
 * 0 = no memory: every render is treated as the first of its view, which is what a camera that moves every frame
 * gets anyway (bench.py reports that time as first_view_ms).
 * MRT_OPT_LISTS (no reference counterpart; default 0): where the tile-culling kernel's tiles get the candidates of their
 * primary rays from -- 0 = by the scene (per-tile lists of primitives from 2 048 primitives on when there are fewer
 * primitives than tiles, lists of 64-primitive leaf clusters when there are more, per-block lists of the loose
 * triangles), 1 = no lists: every tile walks the hierarchy, 2 = lists of primitives and 3 = lists of leaf clusters
 * whatever the scene's size.  Identical buffers in every mode.  MRT_OPT_LIST_CAP (test hook; default 0 = the built-in
 * sizes) shortens every list to that many entries, so that small frames reach the overflow paths (a tile whose list
 * overflowed walks the hierarchy, a block whose list overflowed takes every loose triangle).
 */
enum { MRT_OPT_CULL = 1, MRT_OPT_RNG_SEED = 2, MRT_OPT_FAIL_NEXT_UPLOAD = 3, MRT_OPT_CULL_TEAM = 4, MRT_OPT_VIEW_CACHE = 5,
       MRT_OPT_LISTS = 6, MRT_OPT_LIST_CAP = 7 };
int mrt_ctx_set_option(mrt_ctx *ctx, int option, int value);

/*
 * Host-only diagnostic (no GPU, no context): the conservative bounding ball (centre xyz, radius) the culling
 * kernels use for one triangle given in the 16-float precomputed form.  Radius +inf = never culled.
 */
int mrt_triangle_bound(const float tri16[16], float out4[4]);

/*
 * Host-only diagnostic (no GPU, no context): the XORWOW state curand_init(seed, subsequence, 0, &state) leaves
 * (cuda_utils/curand_state.cuh:22), as the library computes it: out6 = xorshift words v[0..4], Weyl counter d.
 */
int mrt_rng_state(uint64_t seed, uint32_t subsequence, uint32_t out6[6]);

/*
 * Page-locked host memory for frame buffers: replaces the cudaMallocManaged behind rt::ImageBuffer
 * (container/alloc.cuh:30, :72).  mrt_render copies a frame into such a buffer at the speed of the link; pageable
 * memory goes through the runtime's staging copies instead (measured 9 GB/s).  No context needed; memory is
 * usable from every GPU of the process.  mrt_host_free(NULL) is a no-op.
 */
int mrt_host_alloc(size_t bytes, void **out);
int mrt_host_free(void *ptr);

/* --- scene: replaces cudaMallocManaged-backed rt::Vector (container/vector.cuh:26,131) */
int mrt_scene_upload(mrt_ctx *ctx, const mrt_scene_desc *scene);

/* number of local rows a rank renders under a partition (host helper, like
 * rt::make_work_partition, cuda_utils/work_partition.cuh:12-23) */
int mrt_local_rows(int32_t height, const mrt_partition *part, int32_t *rows);

/*
 * Blocking render into HOST buffers: replaces render<<<>>> + cudaGetLastError +
 * cudaDeviceSynchronize (src/main.cu:239-247) and the managed ImageBuffer
 * (container/alloc.cuh:13-106).  Buffers hold the rank's local rows:
 * rgb [rows*W*3] linear float RGB, row 0 = v 0 (the reference's pre-flip order);
 * id [rows*W] object number or -1; t [rows*W] closest-hit parameter or +inf.
 * id/t/stats may be NULL.  params->part.frame_layout must be 0 here.
 */
int mrt_render(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, float *rgb, int32_t *id,
               float *t, mrt_stats *stats);

/*
 * Asynchronous render into DEVICE buffers on a caller-supplied HIP stream
 * (hipStream_t passed as void*; NULL = HIP's default stream, as in HIP itself).  Same layout
 * as mrt_render, or whole frames with params->part.frame_layout = 1.  stats_dev, if not NULL, points
 * to 8 device uint64 that receive primary/shadow/hit/occluded counts, exact-test and opened-cluster counts
 * (2 spare), written by a small kernel that follows the render on the same stream.  The caller synchronises.
 * The first render of a new frame size (small-scene kernel) or of a larger frame (bounce chains) allocates
 * device memory; no call synchronises the device.
 */
int mrt_render_device(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, void *rgb_dev,
                      void *id_dev, void *t_dev, void *stats_dev, void *stream);

/*
 * mrt_render_device for callers that cannot allocate (no counterpart in the reference: it is for code under HIP
 * graph capture, tight serving loops and harnesses that forbid hipMalloc).  Same arguments, same buffers, same
 * results; it calls no hipMalloc / hipFree / allocating routine, ever.  What mrt_render_device keeps in allocated
 * buffers -- the spread ray counters and the small-scene kernel's u / v tables -- lives in one static device array
 * of 144 KB that every GPU the library runs on holds until the process ends.
 * Limits: max_depth 0 only (MRT_ERR_INVALID otherwise; spp 1 or 121), frames of at most 16384 pixels a side
 * (MRT_ERR_TOO_LARGE beyond).  Every call on a GPU -- from any context -- uses that GPU's one array: calls of
 * different contexts that overlap on one GPU race (calls of ONE context are ordered as for mrt_render_device).
 */
int mrt_render_device_static(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, void *rgb_dev,
                             void *id_dev, void *t_dev, void *stats_dev, void *stream);

/*
 * Rank 0 of a multi-GPU frame, after the gather (no reference counterpart: SURVEY 8(e)).  Rank 0 has rendered
 * its own strips in place (frame_layout 1); every other rank r sent ONE packed buffer holding its local rows
 * back to back as three planes, rgb [rows_r*W*3] | id [rows_r*W] | t [rows_r*W] (4-byte elements, rows_r =
 * mrt_local_rows), and staging_dev holds those buffers in rank order, rank skip_rank left out (-1: none is).
 * The call copies every strip of those ranks to its final place in the three H x W frame buffers (frame_id_dev
 * / frame_t_dev may be NULL).  Device pointers; asynchronous on `stream`.  At most 64 ranks.
 */
int mrt_scatter_strips(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev,
                       void *frame_t_dev, int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows,
                       int32_t skip_rank, void *stream);
/*
 * The same when only some planes travel (no reference counterpart).  `planes` = MRT_PLANE_RGB, optionally
 * | MRT_PLANE_ID | MRT_PLANE_T: a packed buffer then holds rgb [rows_r*W*3], then id [rows_r*W] if gathered, then
 * t [rows_r*W] if gathered, and the other ranks render only those (NULL id / t pointers to mrt_render_device) --
 * a caller who wants the picture alone moves 12 bytes per pixel instead of 20.  A plane that does not travel is
 * left as it is in the frame buffers.  mrt_scatter_strips == planes MRT_PLANES_ALL.
 */
enum { MRT_PLANE_RGB = 1, MRT_PLANE_ID = 2, MRT_PLANE_T = 4, MRT_PLANES_ALL = 7 };
int mrt_scatter_strips_planes(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev,
                              void *frame_t_dev, int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows,
                              int32_t skip_rank, int32_t planes, void *stream);

/* ... when the ranks skip_rank .. skip_rank + skip_span - 1 were rendered in place (one context that took several
 * shares of the partition: mrt_partition.rank_span); mrt_scatter_strips_planes == skip_span 1 */
int mrt_scatter_strips_span(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev,
                            void *frame_t_dev, int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows,
                            int32_t skip_rank, int32_t skip_span, int32_t planes, void *stream);

/*
 * --- multi-GPU frames over RCCL, behind the C-ABI (no reference counterpart: the reference is single GPU and has
 * no collective call).  The library opens RCCL (librccl.so.1) at the first mrt_comm_* call; a process
 * that never makes one does not load it.  Two ways to use N GPUs:
 *   one process per GPU   mrt_comm_unique_id on rank 0, the 128 bytes handed to every rank by whatever the caller
 *                         has (a file, a socket, MPI, torch.distributed), then mrt_comm_create on every rank;
 *   one process, N GPUs   mrt_comm_create_local over N contexts (ncclCommInitAll); mrt_render_multi does the rest.
 * A communicator belongs to the context it was created with.  mrt_comm_destroy(NULL) is a no-op.  Destroy order:
 * either may go first -- mrt_comm_destroy never touches the context (it keeps the GPU ordinal itself), and
 * mrt_ctx_destroy does not touch communicators; a communicator whose context is gone can only be destroyed.  Every
 * rank destroys its communicator once its last mrt_gather_strips has completed on the GPU (ncclCommDestroy waits).
 */
typedef struct mrt_comm mrt_comm;
enum { MRT_COMM_ID_BYTES = 128 };
int mrt_comm_unique_id(uint8_t id[MRT_COMM_ID_BYTES]);
/* collective over the num_ranks callers: every rank calls it with the same id */
int mrt_comm_create(mrt_ctx *ctx, const uint8_t id[MRT_COMM_ID_BYTES], int32_t rank, int32_t num_ranks, mrt_comm **out);
/* one process: out[i] is the communicator of ctxs[i], rank i of n (the contexts must sit on n different GPUs) */
int mrt_comm_create_local(mrt_ctx *const *ctxs, int32_t n, mrt_comm **out);
int mrt_comm_destroy(mrt_comm *comm);

/*
 * The one exchange step of a multi-GPU frame, asynchronous on `stream` (a stream of ctx's GPU).  Every rank calls it
 * once per frame with the frame's width, height and strip_rows -- each rank from its own process or thread: RCCL
 * connects two ranks when both have posted, so one thread cannot post for several ranks one after the other
 * (mrt_render_multi is the single-thread form).
 *   rank != 0: sends packed_dev -- its local rows back to back as three planes, rgb [rows*W*3] | id [rows*W] |
 *              t [rows*W], what mrt_render_device leaves with frame_layout 0 in one buffer -- to rank 0 (ncclSend, one
 *              message over the rank's own xGMI link); the other pointers are ignored;
 *   rank 0:    receives every peer's buffer into staging_dev (rank order, one grouped batch of ncclRecv) and then
 *              copies the strips to their places in the three H x W frame planes (mrt_scatter_strips on the same
 *              stream).  With root_in_place != 0 rank 0 has rendered its own strips straight into the frame
 *              (frame_layout 1) and packed_dev is ignored; with 0 its own packed buffer travels like everybody
 *              else's (a send to itself), which is also how the path can be exercised on a single GPU.
 * staging_dev holds 5 dwords per pixel of every rank that sends.
 */
int mrt_gather_strips(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                      void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                      int32_t root_in_place, void *stream);
/* the same with only `planes` travelling (mrt_scatter_strips_planes has the buffer layout): every rank passes the
 * same mask; staging_dev then holds 3 + [id] + [t] dwords per pixel of every rank that sends */
int mrt_gather_strips_planes(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev,
                             void *frame_rgb_dev, void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height,
                             int32_t strip_rows, int32_t root_in_place, int32_t planes, void *stream);
/*
 * The WEIGHTED exchange (no reference counterpart): rank 0 renders more than 1 / N of the frame.  Where the exchange
 * bounds a multi-GPU frame -- rank 0 receives (N - 1) / N of it over its links while a GPU renders all of it in less
 * time -- the frame is dealt in V = N - 1 + root_span shares (strip s -> share s % V); rank 0 renders shares
 * 0 .. root_span - 1 in one call, in place (mrt_partition {0, V, strip_rows, 1, root_span}), rank r >= 1 renders share
 * root_span + r - 1 ({root_span + r - 1, V, strip_rows, 0, 0}) into its packed buffer, and every rank calls this with
 * the same root_span: the peers send, rank 0 receives into staging_dev (rank order) and scatters.  root_span 1 ==
 * mrt_gather_strips_planes with root_in_place.  At most 64 shares.  The frame is bit-identical to the one-GPU frame
 * whatever the shares.
 */
int mrt_gather_strips_root(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                           void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                           int32_t root_span, int32_t planes, void *stream);

/*
 * One process, n GPUs, blocking, HOST buffers: the multi-GPU form of mrt_render.  ctxs[i] renders the strips of rank
 * i (params->part.strip_rows rows each, dealt round robin; part.rank / num_ranks / frame_layout are ignored), rank 0
 * in place, the others into packed buffers that mrt_gather_strips brings to ctxs[0]; the whole frame is then copied
 * to rgb [H*W*3], id [H*W], t [H*W] (id / t / stats may be NULL).  Every context must hold the same scene.
 * comms = what mrt_comm_create_local returned for ctxs (may be NULL when n == 1).  stats: sums over the ranks,
 * kernel_ms = the slowest rank's render kernel.  The result is bit-identical to mrt_render on one GPU.
 */
int mrt_render_multi(mrt_ctx *const *ctxs, mrt_comm *const *comms, int32_t n, const mrt_camera *cam,
                     const mrt_render_params *params, float *rgb, int32_t *id, float *t, mrt_stats *stats);

/*
 * Batched single-primitive intersection on the GPU: replaces the test kernels
 * hit_sphere / hit_triangle <<<1,1>>> (test/spheretest.cu:6-12, test/triangletest.cu:6-12).
 * prim: 4 floats (cx,cy,cz,r) for a sphere, 16 floats (layout of mrt_scene_desc.tri,
 * one triangle) for a triangle.  rays [n*6] = origin xyz, direction xyz.
 * Closed interval [tmin,tmax] (core/interval.cuh:28-29).  Host pointers; outputs
 * hit [n], t [n], point [n*3], normal [n*3]; entries of non-hits are left zero.
 */
int mrt_intersect(mrt_ctx *ctx, int prim_kind, const float *prim, const float *rays, float tmin, float tmax,
                  int64_t n, uint8_t *hit, float *t, float *point, float *normal);
/*
 * The same with any of rt::Interval's four modes (core/interval.cuh:14-42: each bound closed or open;
 * mrt_intersect == lower_closed = upper_closed = 1, the mode the reference's tests and its render path use).
 */
int mrt_intersect_interval(mrt_ctx *ctx, int prim_kind, const float *prim, const float *rays, float tmin, float tmax,
                           int lower_closed, int upper_closed, int64_t n, uint8_t *hit, float *t, float *point,
                           float *normal);

#ifdef __cplusplus
}
#endif
#endif
