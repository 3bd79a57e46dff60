// mrt_api.cpp -- C-ABI of libmrt.so (declared in include/mrt.h).
//
// Thin layer: owns one GPU context (device, stream, scene SoA buffers in HBM,
// output staging buffers), validates shapes on the host before any launch, and
// turns HIP failures into status codes + an error string.  It never exits and
// never throws; the print-and-exit convention of the reference's
// src/check_error.cu:7-15 is kept by the C++ host layer (checkRtErrors).

#include "mrt.h"
#include "mrt_internal.h"
#include "mrt_scene_build.h"

#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is opened at the first mrt_comm_* call

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

struct mrt_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::string err;

  // scene in HBM (one allocation, SoA streams carved out of it)
  void *scene_mem = nullptr;
  size_t scene_bytes = 0;
  mrt::DevScene sc{};
  bool have_scene = false;
  bool any_diffuse = false; // bounce chains (max_depth > 0) over a diffuse object need the random streams below

  // output staging for the blocking host-buffer entry point
  void *out_mem = nullptr;
  size_t out_bytes = 0;
  unsigned long long *stats_dev = nullptr;
  int cull = 1; // MRT_OPT_CULL value
  int team = 0; // MRT_OPT_CULL_TEAM value
  int lists = 0, list_cap = 0; // MRT_OPT_LISTS, MRT_OPT_LIST_CAP

  // u = col / W, v = row / H tables of the small-scene kernel, one entry per frame size seen, kept until the
  // context goes (a frame size costs (W + H) floats)
  struct UvTables { int W, H, nu; float *mem; };
  std::vector<UvTables> uv;
  // host copy of the triangles of a small scene (16 floats each, device order): the small-scene kernel's
  // launcher derives the camera-dependent culling functions from it for every render
  std::vector<float> small_tri16, small_sph;
  std::vector<int32_t> small_tri_obj, small_sph_obj;
  mrt::SmallViewCache small_view; // last view of the small-scene launcher (reset by an upload)
  bool view_cache = true;         // MRT_OPT_VIEW_CACHE

  // per-pixel XORWOW streams of the diffuse bounce (mrt_rng.hip): jump matrices, and the xorshift words of
  // streams 0..rng_n-1 for seed rng_seed (5 streams of rng_n words), built on first need
  uint32_t *rng_jump = nullptr;
  uint32_t *rng_mem = nullptr;
  uint32_t rng_n = 0;
  uint64_t rng_seed = 1984, rng_built_seed = 0; // cuda_utils/curand_state.cuh:22
  // scratch of the bounce kernel's tile ordering (mrt_bounce.hip), grown on demand
  uint32_t *bounce_scratch = nullptr;
  size_t bounce_scratch_words = 0;
  // per-frame block lists of the loose triangles (mrt_cull.hip), grown on demand; never by the allocation-free entry
  void *loose_scratch = nullptr;
  size_t loose_scratch_bytes = 0;
  bool no_alloc_call = false; // set around mrt_render_device_static's launch

  // Renders of one context share stats_dev and the scratch above, so they are ordered on the GPU: a render on
  // another stream than the last one waits for it (order_after_last).  Buffers a render in flight may still read
  // are never freed under it: outgrown ones are parked here until the next point where the streams are drained.
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  hipEvent_t order_ev = nullptr;
  std::vector<void *> retired;
  size_t retired_bytes = 0;
  bool fail_next_upload = false; // MRT_OPT_FAIL_NEXT_UPLOAD (test hook)
  void *static_scratch = nullptr; // this GPU's copy of the allocation-free entry point's array (or null: see create)
};

// the blocking entry point spreads its ray counters over this many copies (same-address atomics serialise: with
// one copy the default frame's 16 200 waves spent 0.7 ms on them) and adds them up on the host
enum { STATS_SLOTS = 256 };

using namespace mrt::build;

// ---- the allocation-free entry point's memory (mrt_render_device_static, include/mrt.h) ----
// One file-scope device array per GPU the library runs on, held until the process ends: the render kernels' ray
// counters (STATS_SLOTS copies of 8) and the small-scene kernel's u = col / W, v = row / H tables for frames of up
// to STATIC_MAX_DIM pixels a side.  In host code `scratch` is only a stand-in for the symbol: the address on a
// GPU comes from hipGetSymbolAddress with that GPU current (static_scratch_of, called by mrt_ctx_create), and
// only that pointer is handed to kernels and memsets.
enum { STATIC_MAX_DIM = 16384, STATIC_MAX_DEVICES = 64 };
enum : size_t {
  STATIC_STATS_BYTES = (size_t)STATS_SLOTS * 8 * sizeof(unsigned long long),
  STATIC_UTAB_FLOATS = STATIC_MAX_DIM + 160, // mrt::uv_table_stride(STATIC_MAX_DIM)
  STATIC_SCRATCH_BYTES = STATIC_STATS_BYTES + (STATIC_UTAB_FLOATS + STATIC_MAX_DIM) * sizeof(float)
};
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
static void *g_static_scratch[STATIC_MAX_DEVICES]; // per device: each GPU holds its own copy at its own address

// the array's address on `device`, which must be the current device; resolved once per device
static hipError_t static_scratch_of(int device, void **out)
{
  *out = nullptr;
  if (device < 0 || device >= STATIC_MAX_DEVICES) return hipErrorInvalidDevice;
  if (!g_static_scratch[device]) {
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    g_static_scratch[device] = p;
  }
  *out = g_static_scratch[device];
  return hipSuccess;
}

static thread_local std::string g_create_err;

static int fail(mrt_ctx *ctx, int code, const std::string &msg)
{
  try { if (ctx) ctx->err = msg; else g_create_err = msg; } catch (...) {} // never throws through the C-ABI
  return code;
}

// Literal messages go straight into the context's string, whose capacity mrt_ctx_create reserved: no heap on the
// error paths of the allocation-free entry point either.
static int fail(mrt_ctx *ctx, int code, const char *msg)
{
  try { if (ctx) ctx->err.assign(msg); else g_create_err.assign(msg); } catch (...) {}
  return code;
}

static int hip_fail(mrt_ctx *ctx, hipError_t e, const char *what)
{
  char buf[256];
  std::snprintf(buf, sizeof buf, "HIP error: %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return fail(ctx, MRT_ERR_HIP, buf);
}

#define HIP_TRY(ctx, expr)                                   \
  do {                                                       \
    hipError_t e_ = (expr);                                  \
    if (e_ != hipSuccess) return hip_fail((ctx), e_, #expr); \
  } while (0)

// Kernel choice (MRT_OPT_CULL / environment MRT_CULL): 0 brute force (every ray x every primitive, like the
// reference loops), 1 automatic (default), 2 tile culling over the primitive hierarchy, 5 the small-scene kernel
// (at most 64 primitives; larger scenes fall back to 2).  All of them produce identical buffers (tests compare them).
static bool cull_mode_ok(int x) { return x == 0 || x == 1 || x == 2 || x == 5; }

static int use_cull()
{
  static const int v = [] {
    const char *e = getenv("MRT_CULL");
    const int x = e ? atoi(e) : 1;
    return cull_mode_ok(x) ? x : 1;
  }();
  return v;
}

// The next render goes to stream st: if the last one went to another stream, st waits for it (on the GPU).
static hipError_t order_after_last(mrt_ctx *ctx, hipStream_t st)
{
  if (ctx->have_last && ctx->last_stream != st) {
    hipError_t e = hipEventRecord(ctx->order_ev, ctx->last_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(st, ctx->order_ev, 0);
    else { // the caller destroyed that stream (which waits for its work): nothing left to order against
      (void)hipGetLastError();
      e = hipSuccess;
    }
    if (e != hipSuccess) return e;
  }
  ctx->last_stream = st;
  ctx->have_last = true;
  return hipSuccess;
}

// Waits (on the host) for everything this context has enqueued; then nothing reads the parked buffers any more.
static hipError_t drain(mrt_ctx *ctx)
{
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) return e;
  if (ctx->have_last && ctx->last_stream != ctx->stream) {
    e = hipStreamSynchronize(ctx->last_stream);
    if (e != hipSuccess) (void)hipGetLastError(); // stream destroyed by the caller: its work is done
  }
  for (void *p : ctx->retired) (void)hipFree(p);
  ctx->retired.clear();
  ctx->retired_bytes = 0;
  return hipSuccess;
}

static hipError_t retire(mrt_ctx *ctx, void *p, size_t bytes)
{
  if (!p) return hipSuccess;
  try { ctx->retired.push_back(p); } catch (...) { return hipErrorOutOfMemory; }
  ctx->retired_bytes += bytes;
  // a caller that keeps growing its frames: give the memory back once it adds up
  return ctx->retired_bytes > ((size_t)1 << 30) ? drain(ctx) : hipSuccess;
}

// The small-scene kernel reads u = col / W and v = row / H from tables.  A new frame size gets its own pair, built
// on the stream of the render ahead of it; renders of one context are ordered (order_after_last), so every later
// render, on whatever stream, runs after the build.  Nothing is freed or synchronised here.
static hipError_t ensure_uv_tables(mrt_ctx *ctx, int W, int H, hipStream_t st, const float **utab, const float **vtab)
{
  for (const auto &u : ctx->uv)
    if (u.W == W && u.H == H) { *utab = u.mem; *vtab = u.mem + u.nu; return hipSuccess; }
  hipError_t e;
  if (ctx->uv.size() >= 256) { // a caller cycling through frame sizes: start over
    if ((e = drain(ctx)) != hipSuccess) return e;
    for (auto &u : ctx->uv) (void)hipFree(u.mem);
    ctx->uv.clear();
  }
  mrt_ctx::UvTables u{W, H, mrt::uv_table_stride(W), nullptr};
  try { ctx->uv.reserve(ctx->uv.size() + 1); } catch (...) { return hipErrorOutOfMemory; }
  if ((e = hipMalloc((void **)&u.mem, sizeof(float) * ((size_t)u.nu + H))) != hipSuccess) return e;
  if ((e = mrt::launch_uv_table(u.mem, u.nu, W, u.mem + u.nu, H, H, st)) != hipSuccess) { (void)hipFree(u.mem); return e; }
  ctx->uv.push_back(u);
  *utab = u.mem; *vtab = u.mem + u.nu;
  return hipSuccess;
}

// Streams 0..npix-1 (the stream of a pixel is its index in the FULL frame, src/main.cu:133, whatever strip
// partition renders it).  Built on the context's stream after the last render (which may still read the old
// states) and waited for on the host: first need, a larger frame or a new seed only.
static hipError_t ensure_rng_states(mrt_ctx *ctx, size_t npix)
{
  if (ctx->rng_mem && ctx->rng_n >= npix && ctx->rng_built_seed == ctx->rng_seed) return hipSuccess;
  hipError_t e;
  if (!ctx->rng_jump) {
    const size_t bytes = (size_t)mrt::RNG_JUMP_MATRICES * 800 * sizeof(uint32_t);
    if ((e = hipMalloc((void **)&ctx->rng_jump, bytes)) != hipSuccess) return e;
    if ((e = hipMemcpy(ctx->rng_jump, mrt::rng_jump_tables(), bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
  }
  if (ctx->rng_n < npix) {
    if ((e = retire(ctx, ctx->rng_mem, 5 * (size_t)ctx->rng_n * sizeof(uint32_t))) != hipSuccess) return e;
    ctx->rng_mem = nullptr; ctx->rng_n = 0;
    const size_t n = (npix + 1023) / 1024 * 1024;
    if ((e = hipMalloc((void **)&ctx->rng_mem, 5 * n * sizeof(uint32_t))) != hipSuccess) return e;
    ctx->rng_n = (uint32_t)n;
  }
  if ((e = order_after_last(ctx, ctx->stream)) != hipSuccess) return e;
  if ((e = mrt::launch_rng_init(ctx->rng_jump, ctx->rng_mem, ctx->rng_n, ctx->rng_n, ctx->rng_seed, ctx->stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return e;
  ctx->rng_built_seed = ctx->rng_seed;
  return hipSuccess;
}

static mrt::SmallHost small_host(mrt_ctx *ctx)
{
  mrt::SmallHost hs;
  hs.view = ctx->view_cache ? &ctx->small_view : nullptr;
  hs.ns = (int)ctx->small_sph_obj.size(); hs.nt = (int)ctx->small_tri_obj.size();
  hs.sph = ctx->small_sph.data(); hs.sph_obj = ctx->small_sph_obj.data();
  hs.tri16 = ctx->small_tri16.data(); hs.tri_obj = ctx->small_tri_obj.data();
  return hs;
}

static hipError_t launch_by_mode(mrt_ctx *ctx, mrt::RenderArgs &A, hipStream_t st)
{
  hipError_t e;
  if (A.max_depth > 0) { // bounce chains: one kernel
    if (ctx->any_diffuse) {
      if ((e = ensure_rng_states(ctx, (size_t)A.W * A.H)) != hipSuccess) return e;
      uint32_t v[5];
      mrt::rng_seed_state(ctx->rng_seed, v, &A.rng_d);
      A.rng = ctx->rng_mem; A.rng_stride = ctx->rng_n;
    }
    if ((e = order_after_last(ctx, st)) != hipSuccess) return e;
    // MRT_OPT_CULL = 0 keeps the small-scene culling data out of it (every ray walks the hierarchy: cross-check)
    const bool small = (long)A.sc.ns + A.sc.nt <= 64 && ctx->cull != 0;
    const mrt::SmallHost hs = small_host(ctx);
    uint32_t *scratch = nullptr;
    if (small && A.spp > 1) {
      const size_t need = mrt::bounce_scratch_words(A.W, A.local_rows);
      if (ctx->bounce_scratch_words < need) {
        if ((e = retire(ctx, ctx->bounce_scratch, ctx->bounce_scratch_words * sizeof(uint32_t))) != hipSuccess) return e;
        ctx->bounce_scratch = nullptr; ctx->bounce_scratch_words = 0;
        if ((e = hipMalloc((void **)&ctx->bounce_scratch, need * sizeof(uint32_t))) != hipSuccess) return e;
        ctx->bounce_scratch_words = need;
      }
      scratch = ctx->bounce_scratch;
    }
    return mrt::launch_render_bounce(A, small ? &hs : nullptr, scratch, st);
  }
  if ((e = order_after_last(ctx, st)) != hipSuccess) return e;
  int mode = ctx->cull;
  const bool small = (long)A.sc.ns + A.sc.nt <= 64;
  // measured: the small-scene kernel wins where it applies; with the primitive-side hierarchy the
  // tile-culling kernel wins everywhere else
  if (mode == 1) mode = small ? 5 : 2;
  if (mode == 5 && !small) mode = 2;
  switch (mode) {
  case 0: return mrt::launch_render(A, st);
  case 2: {
    void *ls = nullptr;
    const size_t need = ctx->no_alloc_call ? 0 : mrt::loose_scratch_bytes(A);
    if (need) {
      if (ctx->loose_scratch_bytes < need) {
        if ((e = retire(ctx, ctx->loose_scratch, ctx->loose_scratch_bytes)) != hipSuccess) return e;
        ctx->loose_scratch = nullptr; ctx->loose_scratch_bytes = 0;
        if ((e = hipMalloc(&ctx->loose_scratch, need)) != hipSuccess) return e;
        ctx->loose_scratch_bytes = need;
      }
      ls = ctx->loose_scratch;
    }
    return mrt::launch_render_cull(A, ls, st);
  }
  default:
    if (!A.utab && (e = ensure_uv_tables(ctx, A.W, A.H, st, &A.utab, &A.vtab)) != hipSuccess) return e;
    return mrt::launch_render_small(A, small_host(ctx), st);
  }
}

extern "C" {

int mrt_ctx_set_option(mrt_ctx *ctx, int option, int value)
{
  if (!ctx) return MRT_ERR_INVALID;
  if (option == MRT_OPT_CULL) {
    if (!cull_mode_ok(value)) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_CULL takes 0, 1, 2 or 5");
    ctx->cull = value;
    return MRT_OK;
  }
  if (option == MRT_OPT_CULL_TEAM) {
    if (value != 0 && value != 1 && value != 4 && value != 5) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_CULL_TEAM takes 0, 1, 4 or 5");
    ctx->team = value;
    return MRT_OK;
  }
  if (option == MRT_OPT_LISTS) {
    if (value < 0 || value > 3) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_LISTS takes 0, 1, 2 or 3");
    ctx->lists = value;
    return MRT_OK;
  }
  if (option == MRT_OPT_LIST_CAP) {
    if (value < 0 || value > 0x7fff) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_LIST_CAP takes 0 (the built-in sizes) or a number of entries up to 32767");
    ctx->list_cap = value;
    return MRT_OK;
  }
  if (option == MRT_OPT_FAIL_NEXT_UPLOAD) {
    ctx->fail_next_upload = value != 0;
    return MRT_OK;
  }
  if (option == MRT_OPT_VIEW_CACHE) {
    if (value != 0 && value != 1) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_VIEW_CACHE takes 0 or 1");
    ctx->view_cache = value != 0;
    ctx->small_view = mrt::SmallViewCache();
    return MRT_OK;
  }
  if (option == MRT_OPT_RNG_SEED) {
    if (value < 0) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_RNG_SEED takes 0..2^31-1");
    ctx->rng_seed = (uint64_t)value;
    return MRT_OK;
  }
  return fail(ctx, MRT_ERR_INVALID, "unknown option");
}

const char *mrt_status_string(int status)
{
  switch (status) {
  case MRT_OK: return "ok";
  case MRT_ERR_INVALID: return "invalid argument";
  case MRT_ERR_HIP: return "HIP runtime error";
  case MRT_ERR_NO_SCENE: return "no scene uploaded";
  case MRT_ERR_NO_DEVICE: return "no usable GPU";
  case MRT_ERR_TOO_LARGE: return "size beyond limit";
  default: return "unknown status";
  }
}

const char *mrt_last_error(const mrt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int mrt_device_count(int *count)
{
  if (!count) return MRT_ERR_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return hip_fail(nullptr, e, "hipGetDeviceCount"); }
  *count = n;
  return MRT_OK;
}

int mrt_ctx_create(int device, mrt_ctx **out)
{
  if (!out) return MRT_ERR_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(nullptr, MRT_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n) return fail(nullptr, MRT_ERR_INVALID, "device index out of range");
  mrt_ctx *ctx = new (std::nothrow) mrt_ctx();
  if (!ctx) return fail(nullptr, MRT_ERR_INVALID, "out of host memory");
  ctx->device = device;
  ctx->cull = use_cull();
  try { ctx->err.reserve(512); } catch (...) { delete ctx; return fail(nullptr, MRT_ERR_INVALID, "out of host memory"); }
  auto bail = [&](hipError_t err, const char *what) {
    int rc = hip_fail(nullptr, err, what);
    mrt_ctx_destroy(ctx);
    return rc;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipMalloc((void **)&ctx->stats_dev, STATS_SLOTS * 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
  // the allocation-free entry point's array on this GPU; a failure here only disables that entry point
  if (static_scratch_of(device, &ctx->static_scratch) != hipSuccess) { (void)hipGetLastError(); ctx->static_scratch = nullptr; }
  *out = ctx;
  return MRT_OK;
}

int mrt_ctx_destroy(mrt_ctx *ctx)
{
  if (!ctx) return MRT_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)drain(ctx);
  if (ctx->scene_mem) (void)hipFree(ctx->scene_mem);
  for (auto &u : ctx->uv) (void)hipFree(u.mem);
  if (ctx->rng_mem) (void)hipFree(ctx->rng_mem);
  if (ctx->rng_jump) (void)hipFree(ctx->rng_jump);
  if (ctx->bounce_scratch) (void)hipFree(ctx->bounce_scratch);
  if (ctx->loose_scratch) (void)hipFree(ctx->loose_scratch);
  if (ctx->out_mem) (void)hipFree(ctx->out_mem);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MRT_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int mrt_host_alloc(size_t bytes, void **out)
{
  if (!out) return MRT_ERR_INVALID;
  *out = nullptr;
  if (bytes == 0) return MRT_OK;
  hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
  if (e != hipSuccess) { *out = nullptr; return hip_fail(nullptr, e, "hipHostMalloc"); }
  return MRT_OK;
}

int mrt_host_free(void *ptr)
{
  if (!ptr) return MRT_OK;
  hipError_t e = hipHostFree(ptr);
  return e == hipSuccess ? MRT_OK : hip_fail(nullptr, e, "hipHostFree");
}

int mrt_rng_state(uint64_t seed, uint32_t subsequence, uint32_t out6[6])
{
  if (!out6) return MRT_ERR_INVALID;
  mrt::rng_host_state(seed, subsequence, out6);
  return MRT_OK;
}

int mrt_triangle_bound(const float tri16[16], float out4[4])
{
  if (!tri16 || !out4) return MRT_ERR_INVALID;
  tri_bound(tri16, out4);
  return MRT_OK;
}

int mrt_scene_upload(mrt_ctx *ctx, const mrt_scene_desc *s)
{
  if (!ctx || !s) return MRT_ERR_INVALID;
  if (s->num_objects < 0 || s->num_spheres < 0 || s->num_triangles < 0 || s->num_lights < 0)
    return fail(ctx, MRT_ERR_INVALID, "negative count in scene");
  if (s->num_objects > 0 && (!s->mat_kind || !s->mat_rgb)) return fail(ctx, MRT_ERR_INVALID, "missing material arrays");
  if (s->num_spheres > 0 && (!s->sphere_cx || !s->sphere_cy || !s->sphere_cz || !s->sphere_r || !s->sphere_obj))
    return fail(ctx, MRT_ERR_INVALID, "missing sphere arrays");
  if (s->num_triangles > 0 && (!s->tri || !s->tri_obj)) return fail(ctx, MRT_ERR_INVALID, "missing triangle arrays");
  if (s->num_lights > 0 && (!s->light_kind || !s->light_vec || !s->light_rgb))
    return fail(ctx, MRT_ERR_INVALID, "missing light arrays");
  if ((int64_t)s->num_spheres + s->num_triangles > s->num_objects)
    return fail(ctx, MRT_ERR_INVALID, "more primitives than objects");
  // object numbers: in range, strictly ascending per type (a scene lists every object once)
  for (int t = 0; t < 2; ++t) {
    const int32_t *o = t ? s->tri_obj : s->sphere_obj;
    const int n = t ? s->num_triangles : s->num_spheres;
    for (int i = 0; i < n; ++i) {
      if (o[i] < 0 || o[i] >= s->num_objects) return fail(ctx, MRT_ERR_INVALID, "primitive object number out of range");
      if (i && o[i] <= o[i - 1]) return fail(ctx, MRT_ERR_INVALID, "primitive object numbers must ascend");
    }
  }
  for (int i = 0; i < s->num_objects; ++i)
    if (s->mat_kind[i] > MRT_MAT_METALLIC) return fail(ctx, MRT_ERR_INVALID, "unknown material kind");
  for (int i = 0; i < s->num_lights; ++i)
    if (s->light_kind[i] > MRT_LIGHT_POINT) return fail(ctx, MRT_ERR_INVALID, "unknown light kind");

  const size_t ns = s->num_spheres, nt = s->num_triangles, no = s->num_objects, nl = s->num_lights;
  // carve one host image, one device allocation, one copy
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_cx = take(ns * 4), o_cy = take(ns * 4), o_cz = take(ns * 4), o_r = take(ns * 4), o_so = take(ns * 4);
  const size_t o_tri = take(nt * 16 * 4), o_to = take(nt * 4), o_tb = take(nt * 4 * 4), o_trec = take(nt * 64);
  const bool small = ns + nt <= 64;
  const size_t nls = small ? std::min<size_t>(nl, SMALL_MAX_SHADOW_LIGHTS) : 0;
  const size_t o_shfn = take(nls * SH_STRIDE * nt * 4);
  const size_t ncs = (ns + 63) / 64, nct = (nt + 63) / 64;
  const size_t o_cs = take(ncs * 16), o_ct = take(nct * 16);
  const size_t nscs = (ncs + 63) / 64, nsct = (nct + 63) / 64;
  const size_t o_scs = take(nscs * 16), o_sct = take(nsct * 16);
  const size_t o_mk = take(no), o_mrgb = take(no * 12);
  const size_t o_lk = take(nl), o_lv = take(nl * 12), o_lc = take(nl * 12);
  const size_t total = off ? off : 256;

  std::vector<unsigned char> img;
  try { img.assign(total, 0); } catch (...) { return fail(ctx, MRT_ERR_INVALID, "out of host memory"); }
  size_t tri_regular = nt; // triangles at the head of the device order that are neither loose nor unbounded
  // host copy of a small scene (device order), committed to the context together with the device scene
  std::vector<float> new_tri16, new_sph;
  std::vector<int32_t> new_tri_obj, new_sph_obj;
  auto put = [&](size_t o, const void *src, size_t bytes) { if (bytes) std::memcpy(img.data() + o, src, bytes); };
  {
    // spheres, in kd order
    std::vector<uint32_t> order;
    primitive_order(ns, [&](size_t i, float *b) { b[0] = s->sphere_cx[i]; b[1] = s->sphere_cy[i]; b[2] = s->sphere_cz[i];
                                               b[3] = s->sphere_r[i]; }, order);
    float *cx = (float *)(img.data() + o_cx), *cy = (float *)(img.data() + o_cy), *cz = (float *)(img.data() + o_cz);
    float *r = (float *)(img.data() + o_r);
    int32_t *so = (int32_t *)(img.data() + o_so);
    for (size_t i = 0; i < ns; ++i) {
      const uint32_t g = order[i];
      cx[i] = s->sphere_cx[g]; cy[i] = s->sphere_cy[g]; cz[i] = s->sphere_cz[g]; r[i] = s->sphere_r[g];
      so[i] = s->sphere_obj[g];
    }
    try {
      if (small)
        for (size_t i = 0; i < ns; ++i) {
          new_sph.insert(new_sph.end(), {cx[i], cy[i], cz[i], r[i]});
          new_sph_obj.push_back(so[i]);
        }
    } catch (...) { return fail(ctx, MRT_ERR_INVALID, "out of host memory"); }
    // a sphere's own ball must contain it even for a negative radius: the hit test only sees r*r
    std::vector<float> rabs(ns);
    for (size_t i = 0; i < ns; ++i) rabs[i] = std::fabs(r[i]);
    float *cs = (float *)(img.data() + o_cs);
    cluster_bounds(cx, cy, cz, rabs.data(), ns, cs);
    cluster_bounds(cs, cs + ncs, cs + 2 * ncs, cs + 3 * ncs, ncs, (float *)(img.data() + o_scs));
  }
  {
    // triangles: bounds of the accepted regions first, then kd order of those
    std::vector<float> tb(4 * nt);
    for (size_t i = 0; i < nt; ++i) {
      float q[16];
      for (int k = 0; k < 16; ++k) q[k] = s->tri[k * nt + i];
      tri_bound(q, &tb[4 * i]);
    }
    std::vector<uint32_t> order;
    primitive_order(nt, [&](size_t i, float *b) { std::memcpy(b, &tb[4 * i], 16); }, order, &tri_regular);
    float *tri = (float *)(img.data() + o_tri), *bnd = (float *)(img.data() + o_tb), *rec = (float *)(img.data() + o_trec);
    int32_t *to = (int32_t *)(img.data() + o_to);
    for (size_t i = 0; i < nt; ++i) {
      const uint32_t g = order[i];
      for (int k = 0; k < 16; ++k) { const float v = s->tri[k * nt + g]; tri[k * nt + i] = v; rec[16 * i + k] = v; }
      for (int k = 0; k < 4; ++k) bnd[k * nt + i] = tb[4 * g + k];
      to[i] = s->tri_obj[g];
    }
    // small scenes: shadow-ray functions per (light, triangle), and a host copy of the triangles
    float *shfn = (float *)(img.data() + o_shfn);
    for (size_t l = 0; l < nls; ++l)
      for (size_t i = 0; i < nt; ++i) {
        float f[SH_STRIDE];
        shadow_functions(rec + 16 * i, s->light_kind[l], s->light_vec + 3 * l, f);
        for (int k = 0; k < SH_STRIDE; ++k) shfn[(l * SH_STRIDE + k) * nt + i] = f[k];
      }
    try {
      if (small) { new_tri16.assign(rec, rec + 16 * nt); new_tri_obj.assign(to, to + nt); }
    } catch (...) { return fail(ctx, MRT_ERR_INVALID, "out of host memory"); }
    float *ct = (float *)(img.data() + o_ct);
    cluster_bounds(bnd, bnd + nt, bnd + 2 * nt, bnd + 3 * nt, nt, ct);
    cluster_bounds(ct, ct + nct, ct + 2 * nct, ct + 3 * nct, nct, (float *)(img.data() + o_sct));
  }
  put(o_mk, s->mat_kind, no);      put(o_mrgb, s->mat_rgb, no * 12);
  put(o_lk, s->light_kind, nl);    put(o_lv, s->light_vec, nl * 12); put(o_lc, s->light_rgb, nl * 12);

  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, drain(ctx)); // renders in flight still read the old scene
  // from here on the old scene is given up: whatever fails below leaves the context without one
  ctx->have_scene = false;
  ctx->any_diffuse = false;
  ctx->small_sph.clear(); ctx->small_sph_obj.clear(); ctx->small_tri16.clear(); ctx->small_tri_obj.clear();
  ctx->small_view = mrt::SmallViewCache();
  if (ctx->fail_next_upload) {
    ctx->fail_next_upload = false;
    if (ctx->scene_mem) { (void)hipFree(ctx->scene_mem); ctx->scene_mem = nullptr; ctx->scene_bytes = 0; }
    return fail(ctx, MRT_ERR_HIP, "HIP error: injected allocation failure (MRT_OPT_FAIL_NEXT_UPLOAD) in mrt_scene_upload");
  }
  if (ctx->scene_bytes < total) {
    if (ctx->scene_mem) { void *old = ctx->scene_mem; ctx->scene_mem = nullptr; ctx->scene_bytes = 0; HIP_TRY(ctx, hipFree(old)); }
    HIP_TRY(ctx, hipMalloc(&ctx->scene_mem, total));
    ctx->scene_bytes = total;
  }
  HIP_TRY(ctx, hipMemcpy(ctx->scene_mem, img.data(), total, hipMemcpyHostToDevice));

  unsigned char *b = (unsigned char *)ctx->scene_mem;
  mrt::DevScene &d = ctx->sc;
  d.sph_cx = (const float *)(b + o_cx); d.sph_cy = (const float *)(b + o_cy); d.sph_cz = (const float *)(b + o_cz);
  d.sph_r = (const float *)(b + o_r);   d.sph_obj = (const int32_t *)(b + o_so); d.ns = (int32_t)ns;
  d.tri = (const float *)(b + o_tri);   d.tri_obj = (const int32_t *)(b + o_to); d.nt = (int32_t)nt;
  d.tri_bx = (const float *)(b + o_tb); d.tri_by = d.tri_bx + nt; d.tri_bz = d.tri_bx + 2 * nt; d.tri_br = d.tri_bx + 3 * nt;
  d.tri_rec = (const float *)(b + o_trec);
  d.tri_shfn = (const float *)(b + o_shfn); d.nls = (int32_t)nls;
  d.sph_clu = (const float *)(b + o_cs); d.ncs = (int32_t)ncs;
  d.tri_clu = (const float *)(b + o_ct); d.nct = (int32_t)nct;
  d.nct_regular = (int32_t)std::min(nct, (tri_regular + 63) / 64); // clusters from here on hold loose triangles only
  d.sph_sclu = (const float *)(b + o_scs); d.nscs = (int32_t)nscs;
  d.tri_sclu = (const float *)(b + o_sct); d.nsct = (int32_t)nsct;
  d.mat_kind = b + o_mk; d.mat_rgb = (const float *)(b + o_mrgb); d.nobj = (int32_t)no;
  d.light_kind = b + o_lk; d.light_vec = (const float *)(b + o_lv); d.light_rgb = (const float *)(b + o_lc);
  d.nl = (int32_t)nl;
  for (size_t i = 0; i < no; ++i) ctx->any_diffuse |= (s->mat_kind[i] == MRT_MAT_DIFFUSE);
  ctx->small_sph.swap(new_sph); ctx->small_sph_obj.swap(new_sph_obj);
  ctx->small_tri16.swap(new_tri16); ctx->small_tri_obj.swap(new_tri_obj);
  ctx->small_view = mrt::SmallViewCache();
  ctx->have_scene = true;
  return MRT_OK;
}

static int span_of(const mrt_partition &p) { return p.rank_span > 1 ? p.rank_span : 1; }

static bool partition_ok(const mrt_partition &p)
{
  return p.num_ranks >= 1 && p.num_ranks <= 0xffff && p.rank >= 0 && p.rank_span >= 0 && p.rank + span_of(p) <= p.num_ranks &&
         p.strip_rows >= 8 && (p.strip_rows % 8) == 0 && (p.frame_layout == 0 || p.frame_layout == 1);
}

static int local_rows_of(int H, const mrt_partition &p)
{
  const int nstrips = (H + p.strip_rows - 1) / p.strip_rows, span = span_of(p);
  int rows = 0;
  for (int s = 0; s < nstrips; ++s) {
    const int r = s % p.num_ranks;
    if (r < p.rank || r >= p.rank + span) continue;
    const int r0 = s * p.strip_rows;
    rows += (r0 + p.strip_rows <= H) ? p.strip_rows : (H - r0);
  }
  return rows;
}

int mrt_local_rows(int32_t height, const mrt_partition *part, int32_t *rows)
{
  if (!part || !rows || height < 0 || !partition_ok(*part)) return MRT_ERR_INVALID;
  *rows = local_rows_of(height, *part);
  return MRT_OK;
}

static int fill_args(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *p, mrt::RenderArgs &A)
{
  if (!ctx || !cam || !p) return MRT_ERR_INVALID;
  if (!ctx->have_scene) return fail(ctx, MRT_ERR_NO_SCENE, "mrt_render before mrt_scene_upload");
  if (p->width <= 0 || p->height <= 0) return fail(ctx, MRT_ERR_INVALID, "non-positive frame size");
  if (p->width > 65536 || p->height > 65536) return fail(ctx, MRT_ERR_TOO_LARGE, "frame larger than 65536 in a dimension");
  if (p->spp != 1 && p->spp != 121) return fail(ctx, MRT_ERR_INVALID, "spp must be 1 or 121");
  if (p->max_depth < 0 || p->max_depth > 8) return fail(ctx, MRT_ERR_INVALID, "max_depth must be 0..8");
  if (p->max_depth > 0 && ctx->any_diffuse && (uint64_t)p->width * (uint64_t)p->height > ((uint64_t)1 << 28))
    return fail(ctx, MRT_ERR_TOO_LARGE, "diffuse bounces: more than 2^28 pixels (one random stream of 20 bytes per pixel)");
  if (!partition_ok(p->part)) return fail(ctx, MRT_ERR_INVALID, "bad strip partition");
  std::memset(&A, 0, sizeof A);
  A.sc = ctx->sc;
  A.cam.ox = cam->origin[0]; A.cam.oy = cam->origin[1]; A.cam.oz = cam->origin[2];
  A.cam.hx = cam->horizontal[0]; A.cam.hy = cam->horizontal[1]; A.cam.hz = cam->horizontal[2];
  A.cam.vx = cam->vertical[0]; A.cam.vy = cam->vertical[1]; A.cam.vz = cam->vertical[2];
  A.cam.ux = cam->up_left[0]; A.cam.uy = cam->up_left[1]; A.cam.uz = cam->up_left[2];
  A.W = p->width; A.H = p->height;
  A.rank = mrt::pack_rank(p->part.rank, span_of(p->part)); A.nranks = p->part.num_ranks; A.strip_rows = p->part.strip_rows;
  A.local_rows = local_rows_of(p->height, p->part);
  A.frame_layout = p->part.frame_layout;
  A.spp = p->spp;
  A.max_depth = p->max_depth;
  A.stats_slots = 1;
  A.team_mode = ctx->team | (ctx->lists << 8) | (ctx->list_cap << 16); // opt_team / opt_lists / opt_list_cap
  return MRT_OK;
}

int mrt_render_device(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, void *rgb_dev,
                      void *id_dev, void *t_dev, void *stats_dev, void *stream)
{
  mrt::RenderArgs A;
  int rc = fill_args(ctx, cam, params, A);
  if (rc != MRT_OK) return rc;
  if (!rgb_dev) return fail(ctx, MRT_ERR_INVALID, "rgb buffer is required");
  A.rgb = (float *)rgb_dev; A.id = (int32_t *)id_dev; A.t = (float *)t_dev;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream; // NULL = HIP's default stream, as everywhere in HIP
  HIP_TRY(ctx, order_after_last(ctx, st)); // before anything touches the shared counters
  if (stats_dev) { // counted in the context's spread copies, then added up into the caller's eight words
    A.stats = ctx->stats_dev;
    A.stats_slots = STATS_SLOTS;
    HIP_TRY(ctx, hipMemsetAsync(A.stats, 0, STATS_SLOTS * 8 * sizeof(unsigned long long), st));
  }
  HIP_TRY(ctx, launch_by_mode(ctx, A, st));
  if (stats_dev) HIP_TRY(ctx, mrt::launch_stats_reduce(ctx->stats_dev, STATS_SLOTS, (unsigned long long *)stats_dev, st));
  return MRT_OK;
}

int mrt_render_device_static(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, void *rgb_dev,
                             void *id_dev, void *t_dev, void *stats_dev, void *stream)
{
  mrt::RenderArgs A;
  int rc = fill_args(ctx, cam, params, A);
  if (rc != MRT_OK) return rc;
  if (!rgb_dev) return fail(ctx, MRT_ERR_INVALID, "rgb buffer is required");
  if (A.max_depth != 0)
    return fail(ctx, MRT_ERR_INVALID, "mrt_render_device_static renders max_depth 0 only (bounce chains need buffers sized by the frame)");
  if (A.W > STATIC_MAX_DIM || A.H > STATIC_MAX_DIM)
    return fail(ctx, MRT_ERR_TOO_LARGE, "mrt_render_device_static: frame larger than 16384 pixels in a dimension");
  if (!ctx->static_scratch) return fail(ctx, MRT_ERR_HIP, "mrt_render_device_static: the static device array could not be resolved on this GPU");
  A.rgb = (float *)rgb_dev; A.id = (int32_t *)id_dev; A.t = (float *)t_dev;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  unsigned char *sp = (unsigned char *)ctx->static_scratch;
  unsigned long long *slots = (unsigned long long *)sp;
  float *utab = (float *)(sp + STATIC_STATS_BYTES), *vtab = utab + mrt::uv_table_stride(A.W); // <= STATIC_UTAB_FLOATS
  HIP_TRY(ctx, order_after_last(ctx, st));
  if (stats_dev) { // the array keeps the last call's counts: clear them
    A.stats = slots;
    A.stats_slots = STATS_SLOTS;
    HIP_TRY(ctx, hipMemsetAsync(slots, 0, STATIC_STATS_BYTES, st));
  }
  if ((long)A.sc.ns + A.sc.nt <= 64 && (ctx->cull == 1 || ctx->cull == 5)) {
    // the small-scene kernel's tables: rebuilt on every call, on the caller's stream (the array is shared by every
    // context on this GPU, so nothing is known about what the last call left in it)
    HIP_TRY(ctx, mrt::launch_uv_table(utab, mrt::uv_table_stride(A.W), A.W, vtab, A.H, A.H, st));
    A.utab = utab; A.vtab = vtab;
  }
  ctx->no_alloc_call = true; // nothing below may allocate: the kernels run without their optional scratch buffers
  const hipError_t le = launch_by_mode(ctx, A, st);
  ctx->no_alloc_call = false;
  HIP_TRY(ctx, le);
  if (stats_dev) HIP_TRY(ctx, mrt::launch_stats_reduce(slots, STATS_SLOTS, (unsigned long long *)stats_dev, st));
  return MRT_OK;
}

int mrt_render(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, float *rgb, int32_t *id, float *t,
               mrt_stats *stats)
{
  mrt::RenderArgs A;
  int rc = fill_args(ctx, cam, params, A);
  if (rc != MRT_OK) return rc;
  if (!rgb) return fail(ctx, MRT_ERR_INVALID, "rgb buffer is required");
  if (A.frame_layout != 0) return fail(ctx, MRT_ERR_INVALID, "mrt_render takes frame_layout 0 (a rank's rows back to back)");
  const size_t npix = (size_t)A.local_rows * A.W;
  const size_t need = npix * 20 + 256;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->out_bytes < need) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->out_mem) { HIP_TRY(ctx, hipFree(ctx->out_mem)); ctx->out_mem = nullptr; ctx->out_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->out_mem, need));
    ctx->out_bytes = need;
  }
  A.rgb = (float *)ctx->out_mem;
  A.id = (int32_t *)((char *)ctx->out_mem + npix * 12);
  A.t = (float *)((char *)ctx->out_mem + npix * 16);
  A.stats = stats ? ctx->stats_dev : nullptr;
  A.stats_slots = STATS_SLOTS;
  HIP_TRY(ctx, order_after_last(ctx, ctx->stream));
  if (A.stats) HIP_TRY(ctx, hipMemsetAsync(A.stats, 0, STATS_SLOTS * 8 * sizeof(unsigned long long), ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  HIP_TRY(ctx, launch_by_mode(ctx, A, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  if (npix) {
    HIP_TRY(ctx, hipMemcpyAsync(rgb, A.rgb, npix * 12, hipMemcpyDeviceToHost, ctx->stream));
    if (id) HIP_TRY(ctx, hipMemcpyAsync(id, A.id, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (t) HIP_TRY(ctx, hipMemcpyAsync(t, A.t, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
  }
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<unsigned long long> slots;
  if (stats) {
    slots.resize((size_t)STATS_SLOTS * 8);
    HIP_TRY(ctx, hipMemcpyAsync(slots.data(), ctx->stats_dev, slots.size() * sizeof(unsigned long long),
                                hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (stats) {
    for (int s = 0; s < STATS_SLOTS; ++s)
      for (int k = 0; k < 8; ++k) h[k] += slots[(size_t)s * 8 + k];
    stats->primary_rays = h[0]; stats->shadow_rays = h[1]; stats->hit_pixels = h[2]; stats->occluded = h[3];
    stats->exact_tests = h[4]; stats->clusters_opened = h[5];
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->kernel_ms = ms;
  }
  return MRT_OK;
}

static bool planes_ok(int32_t planes) { return (planes & MRT_PLANE_RGB) && (planes & ~MRT_PLANES_ALL) == 0; }
static int plane_dwords(int32_t planes) { return 3 + ((planes & MRT_PLANE_ID) ? 1 : 0) + ((planes & MRT_PLANE_T) ? 1 : 0); }

int mrt_scatter_strips(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev, void *frame_t_dev,
                       int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows, int32_t skip_rank, void *stream)
{
  return mrt_scatter_strips_planes(ctx, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height, num_ranks,
                                   strip_rows, skip_rank, MRT_PLANES_ALL, stream);
}

int mrt_scatter_strips_planes(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev,
                              void *frame_t_dev, int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows,
                              int32_t skip_rank, int32_t planes, void *stream)
{
  return mrt_scatter_strips_span(ctx, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height, num_ranks,
                                 strip_rows, skip_rank, 1, planes, stream);
}

int mrt_scatter_strips_span(mrt_ctx *ctx, const void *staging_dev, void *frame_rgb_dev, void *frame_id_dev,
                            void *frame_t_dev, int32_t width, int32_t height, int32_t num_ranks, int32_t strip_rows,
                            int32_t skip_rank, int32_t skip_span, int32_t planes, void *stream)
{
  if (!ctx) return MRT_ERR_INVALID;
  if (skip_span < 1 || (skip_rank >= 0 && skip_rank + skip_span > num_ranks))
    return fail(ctx, MRT_ERR_INVALID, "skip_span out of range");
  if (!planes_ok(planes)) return fail(ctx, MRT_ERR_INVALID, "planes: MRT_PLANE_RGB, optionally | MRT_PLANE_ID | MRT_PLANE_T");
  if (!staging_dev || !frame_rgb_dev) return fail(ctx, MRT_ERR_INVALID, "staging and rgb frame buffers are required");
  if (width <= 0 || height <= 0) return fail(ctx, MRT_ERR_INVALID, "non-positive frame size");
  if (num_ranks > mrt::MAX_RANKS) return fail(ctx, MRT_ERR_TOO_LARGE, "more than 64 ranks");
  if (skip_rank < -1 || skip_rank >= num_ranks) return fail(ctx, MRT_ERR_INVALID, "skip_rank out of range");
  mrt_partition p{0, num_ranks, strip_rows, 0};
  if (!partition_ok(p)) return fail(ctx, MRT_ERR_INVALID, "bad strip partition");
  mrt::ScatterArgs S;
  std::memset(&S, 0, sizeof S);
  S.staging = (const uint32_t *)staging_dev;
  S.rgb = (uint32_t *)frame_rgb_dev; S.id = (uint32_t *)frame_id_dev; S.t = (uint32_t *)frame_t_dev;
  S.W = width; S.H = height; S.nranks = num_ranks; S.strip_rows = strip_rows;
  S.skip_rank = skip_rank < 0 ? num_ranks : skip_rank; S.skip_span = skip_rank < 0 ? 0 : skip_span;
  S.has_id = (planes & MRT_PLANE_ID) ? 1 : 0; S.has_t = (planes & MRT_PLANE_T) ? 1 : 0;
  uint64_t off = 0;
  for (int r = 0; r < num_ranks; ++r) {
    p.rank = r;
    S.rank_rows[r] = local_rows_of(height, p);
    S.rank_off[r] = off;
    const bool skipped = skip_rank >= 0 && r >= skip_rank && r < skip_rank + skip_span;
    if (!skipped) off += (uint64_t)S.rank_rows[r] * (uint64_t)width * (uint64_t)plane_dwords(planes);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, mrt::launch_scatter(S, (hipStream_t)stream));
  return MRT_OK;
}

int mrt_intersect(mrt_ctx *ctx, int prim_kind, const float *prim, const float *rays, float tmin, float tmax, int64_t n,
                  uint8_t *hit, float *t, float *point, float *normal)
{
  return mrt_intersect_interval(ctx, prim_kind, prim, rays, tmin, tmax, 1, 1, n, hit, t, point, normal);
}

int mrt_intersect_interval(mrt_ctx *ctx, int prim_kind, const float *prim, const float *rays, float tmin, float tmax,
                           int lower_closed, int upper_closed, int64_t n, uint8_t *hit, float *t, float *point,
                           float *normal)
{
  const int mode = (lower_closed ? 0 : 1) | (upper_closed ? 0 : 2);
  if (!ctx || !prim || n < 0 || (n > 0 && (!rays || !hit || !t || !point || !normal))) return MRT_ERR_INVALID;
  if (prim_kind != MRT_GEOM_SPHERE && prim_kind != MRT_GEOM_PLANE && prim_kind != MRT_GEOM_TRIANGLE)
    return fail(ctx, MRT_ERR_INVALID, "unknown primitive kind");
  if (n == 0) return MRT_OK;
  if (n > (int64_t)1 << 28) return fail(ctx, MRT_ERR_TOO_LARGE, "more than 2^28 rays in one call");
  const size_t nprim = (prim_kind == MRT_GEOM_TRIANGLE ? 16 : 4) * sizeof(float);
  const size_t b_rays = (size_t)n * 24, b_hit = align_up((size_t)n, 256), b_t = (size_t)n * 4, b_v = (size_t)n * 12;
  const size_t total = 256 + b_rays + b_hit + b_t + 2 * b_v;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  char *mem = nullptr;
  HIP_TRY(ctx, hipMalloc((void **)&mem, total));
  float *d_prim = (float *)mem;
  float *d_rays = (float *)(mem + 256);
  float *d_t = (float *)(mem + 256 + b_rays);
  float *d_point = (float *)(mem + 256 + b_rays + b_t);
  float *d_normal = (float *)(mem + 256 + b_rays + b_t + b_v);
  uint8_t *d_hit = (uint8_t *)(mem + 256 + b_rays + b_t + 2 * b_v);
  int rc = MRT_OK;
  hipError_t e;
  do {
    if ((e = hipMemcpyAsync(d_prim, prim, nprim, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(d_rays, rays, b_rays, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
    if ((e = mrt::launch_intersect(prim_kind, d_prim, d_rays, tmin, tmax, mode, n, d_hit, d_t, d_point, d_normal,
                                   ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(hit, d_hit, (size_t)n, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(t, d_t, b_t, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(point, d_point, b_v, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(normal, d_normal, b_v, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) break;
    e = hipStreamSynchronize(ctx->stream);
  } while (0);
  if (e != hipSuccess) rc = hip_fail(ctx, e, "mrt_intersect");
  (void)hipFree(mem);
  return rc;
}

} // extern "C"

// ---- multi-GPU: RCCL behind the C-ABI (include/mrt.h; no reference counterpart) ----------------------
// RCCL is opened on first use, by soname: a process that has loaded a librccl.so.1 already (torch ships one) gets
// that one, everybody else the one of the ROCm installation.  libmrt.so itself does not depend on it.
namespace {
struct Rccl {
  void *lib = nullptr;
  std::string err;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl *rccl(std::string *why)
{
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, [] {
    // MRT_RCCL_LIB names a library outright.  Otherwise an RCCL the process has loaded already is preferred (torch
    // ships one and registers it as "librccl.so": sharing it is what a second torch communicator would do, whereas a
    // second RCCL instance beside it doubles the proxy threads and buffers), then the ROCm installation's.
    const char *env = getenv("MRT_RCCL_LIB");
    if (env && *env) {
      R.lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
      if (!R.lib) R.err = dlerror();
    } else {
      for (const char *n : {"librccl.so", "librccl.so.1"})
        if (!R.lib) R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
      for (const char *n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (R.lib) break;
        R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!R.lib) R.err = dlerror();
      }
    }
    if (!R.lib) { R.err = "cannot open RCCL (librccl.so.1): " + R.err; return; }
    bool ok = true;
    auto sym = [&](const char *name) { void *f = dlsym(R.lib, name); if (!f) { ok = false; R.err = std::string("RCCL lacks ") + name; } return f; };
    R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
    R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
    R.CommInitAll = (decltype(R.CommInitAll))sym("ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
    R.Send = (decltype(R.Send))sym("ncclSend");
    R.Recv = (decltype(R.Recv))sym("ncclRecv");
    R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
    R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(R.lib); R.lib = nullptr; }
  });
  if (!R.lib) { if (why) *why = R.err; return nullptr; }
  return &R;
}

int rccl_fail(mrt_ctx *ctx, const Rccl *R, ncclResult_t r, const char *what)
{
  char buf[256];
  std::snprintf(buf, sizeof buf, "RCCL error: %d (%s) in %s", (int)r, R ? R->GetErrorString(r) : "?", what);
  return fail(ctx, MRT_ERR_HIP, buf);
}
} // namespace

struct mrt_comm {
  ncclComm_t comm = nullptr;
  mrt_ctx *ctx = nullptr; // identity check in mrt_gather_strips only after creation; never dereferenced by destroy
  int device = 0;         // the context's GPU, kept here so that mrt_comm_destroy works whichever is destroyed first
  int rank = 0, nranks = 1;
};

#define RCCL_TRY(ctx, R, expr)                                        \
  do {                                                                \
    ncclResult_t r_ = (expr);                                         \
    if (r_ != ncclSuccess) return rccl_fail((ctx), (R), r_, #expr);   \
  } while (0)

// dword offset of every rank's packed buffer in rank 0's staging area, and its length; `skip` sends nothing
static uint64_t staging_layout(int W, int H, int nranks, int strip_rows, int skip, int dwords_per_pixel, uint64_t *off,
                               uint64_t *len)
{
  uint64_t o = 0;
  mrt_partition p{0, nranks, strip_rows, 0};
  for (int r = 0; r < nranks; ++r) {
    p.rank = r;
    off[r] = o;
    len[r] = r == skip ? 0 : (uint64_t)local_rows_of(H, p) * (uint64_t)W * (uint64_t)dwords_per_pixel;
    o += len[r];
  }
  return o;
}

extern "C" {

int mrt_comm_unique_id(uint8_t id[MRT_COMM_ID_BYTES])
{
  if (!id) return MRT_ERR_INVALID;
  std::string why;
  const Rccl *R = rccl(&why);
  if (!R) return fail(nullptr, MRT_ERR_HIP, why);
  static_assert(sizeof(ncclUniqueId) == MRT_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  RCCL_TRY(nullptr, R, R->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof u);
  return MRT_OK;
}

int mrt_comm_create(mrt_ctx *ctx, const uint8_t id[MRT_COMM_ID_BYTES], int32_t rank, int32_t num_ranks, mrt_comm **out)
{
  if (!ctx || !id || !out) return MRT_ERR_INVALID;
  *out = nullptr;
  if (num_ranks < 1 || num_ranks > mrt::MAX_RANKS || rank < 0 || rank >= num_ranks)
    return fail(ctx, MRT_ERR_INVALID, "mrt_comm_create: rank / num_ranks out of range (at most 64 ranks)");
  std::string why;
  const Rccl *R = rccl(&why);
  if (!R) return fail(ctx, MRT_ERR_HIP, why);
  mrt_comm *c = new (std::nothrow) mrt_comm();
  if (!c) return fail(ctx, MRT_ERR_INVALID, "out of host memory");
  c->ctx = ctx; c->device = ctx->device; c->rank = rank; c->nranks = num_ranks;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) { delete c; return hip_fail(ctx, e, "hipSetDevice"); }
  ncclResult_t r = R->CommInitRank(&c->comm, num_ranks, u, rank);
  if (r != ncclSuccess) { delete c; return rccl_fail(ctx, R, r, "ncclCommInitRank"); }
  *out = c;
  return MRT_OK;
}

int mrt_comm_create_local(mrt_ctx *const *ctxs, int32_t n, mrt_comm **out)
{
  if (!ctxs || !out || n < 1 || n > mrt::MAX_RANKS) return MRT_ERR_INVALID;
  for (int i = 0; i < n; ++i) { out[i] = nullptr; if (!ctxs[i]) return MRT_ERR_INVALID; }
  mrt_ctx *c0 = ctxs[0];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (ctxs[i]->device == ctxs[j]->device) return fail(c0, MRT_ERR_INVALID, "mrt_comm_create_local: two contexts on one GPU");
  std::string why;
  const Rccl *R = rccl(&why);
  if (!R) return fail(c0, MRT_ERR_HIP, why);
  int devs[mrt::MAX_RANKS];
  ncclComm_t comms[mrt::MAX_RANKS];
  for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
  RCCL_TRY(c0, R, R->CommInitAll(comms, n, devs));
  for (int i = 0; i < n; ++i) {
    mrt_comm *c = new (std::nothrow) mrt_comm();
    if (!c) {
      for (int j = 0; j < n; ++j) { (void)R->CommDestroy(comms[j]); delete out[j]; out[j] = nullptr; }
      return fail(c0, MRT_ERR_INVALID, "out of host memory");
    }
    c->comm = comms[i]; c->ctx = ctxs[i]; c->device = ctxs[i]->device; c->rank = i; c->nranks = n;
    out[i] = c;
  }
  return MRT_OK;
}

int mrt_comm_destroy(mrt_comm *comm)
{
  if (!comm) return MRT_OK;
  const Rccl *R = rccl(nullptr);
  int rc = MRT_OK;
  if (R && comm->comm) {
    (void)hipSetDevice(comm->device);
    ncclResult_t r = R->CommDestroy(comm->comm);
    // the context may be gone already (either destroy order is allowed, include/mrt.h): the message goes to the
    // thread's creation-error slot, which mrt_last_error(NULL) reads
    if (r != ncclSuccess) rc = rccl_fail(nullptr, R, r, "ncclCommDestroy");
  }
  delete comm;
  return rc;
}

// scatter = false: the caller has a group open (the receives are only enqueued when it closes) and scatters itself.
// root_span: rank 0 has rendered the strips of that many shares of a partition into root_span + N - 1 (mrt_partition
// {0, V, strip_rows, 1, root_span}) in place and rank r >= 1 sends the rows of share root_span + r - 1; 1 = the plain
// partition into N.  Shares (the partition's "ranks") and real ranks only differ when root_span > 1.
static int gather_strips(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                         void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                         int32_t root_in_place, int32_t root_span, int32_t planes, void *stream, bool scatter)
{
  if (!ctx || !comm) return MRT_ERR_INVALID;
  if (!planes_ok(planes)) return fail(ctx, MRT_ERR_INVALID, "planes: MRT_PLANE_RGB, optionally | MRT_PLANE_ID | MRT_PLANE_T");
  if (comm->ctx != ctx) return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: the communicator belongs to another context");
  if (width <= 0 || height <= 0) return fail(ctx, MRT_ERR_INVALID, "non-positive frame size");
  const int N = comm->nranks, me = comm->rank;
  if (root_span < 1 || (root_span > 1 && !root_in_place))
    return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: root_span >= 1, and more than one share only with rank 0 rendering in place");
  const int V = N - 1 + root_span;             // shares of the partition
  if (V > mrt::MAX_RANKS) return fail(ctx, MRT_ERR_TOO_LARGE, "more than 64 shares (ranks - 1 + root_span)");
  auto share_of = [&](int r) { return r == 0 ? 0 : root_span + r - 1; }; // a real rank's (first) share
  mrt_partition p{share_of(me), V, strip_rows, 0, me == 0 ? root_span : 1};
  if (!partition_ok(p)) return fail(ctx, MRT_ERR_INVALID, "bad strip partition");
  const Rccl *R = rccl(nullptr);
  if (!R) return fail(ctx, MRT_ERR_HIP, "RCCL is not open");
  // per REAL rank: where its packed buffer lands in rank 0's staging area (rank order), and its length
  uint64_t off[mrt::MAX_RANKS], len[mrt::MAX_RANKS];
  {
    uint64_t o = 0;
    for (int r = 0; r < N; ++r) {
      mrt_partition q{share_of(r), V, strip_rows, 0, r == 0 ? root_span : 1};
      off[r] = o;
      len[r] = (r == 0 && root_in_place) ? 0 : (uint64_t)local_rows_of(height, q) * (uint64_t)width * (uint64_t)plane_dwords(planes);
      o += len[r];
    }
  }
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (me != 0) {
    if (len[me] == 0) return MRT_OK; // more ranks than strips: nothing to send (rank 0 posts no receive either)
    if (!packed_dev) return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: packed buffer is required on ranks other than 0");
    RCCL_TRY(ctx, R, R->Send(packed_dev, (size_t)len[me], ncclFloat, 0, comm->comm, st));
    return MRT_OK;
  }
  if (!frame_rgb_dev) return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: rank 0 needs the frame buffers");
  bool any = false;
  for (int r = 0; r < N; ++r) any |= len[r] != 0;
  if (!any) return MRT_OK; // one rank, rendered in place
  if (!staging_dev) return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: rank 0 needs the staging buffer");
  if (!root_in_place && len[0] && !packed_dev) return fail(ctx, MRT_ERR_INVALID, "mrt_gather_strips: rank 0's packed buffer is required without root_in_place");
  // one grouped batch: every peer's message lands at its offset (and rank 0's own, sent to itself, when asked)
  RCCL_TRY(ctx, R, R->GroupStart());
  ncclResult_t r1 = ncclSuccess;
  if (!root_in_place && len[0]) r1 = R->Send(packed_dev, (size_t)len[0], ncclFloat, 0, comm->comm, st);
  for (int r = 0; r < N && r1 == ncclSuccess; ++r)
    if (len[r]) r1 = R->Recv((uint32_t *)staging_dev + off[r], (size_t)len[r], ncclFloat, r, comm->comm, st);
  ncclResult_t r2 = R->GroupEnd();
  if (r1 != ncclSuccess) return rccl_fail(ctx, R, r1, "ncclSend / ncclRecv");
  if (r2 != ncclSuccess) return rccl_fail(ctx, R, r2, "ncclGroupEnd");
  if (!scatter) return MRT_OK;
  return mrt_scatter_strips_span(ctx, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height, V, strip_rows,
                                 root_in_place ? 0 : -1, root_span, planes, stream);
}

int mrt_gather_strips(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                      void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                      int32_t root_in_place, void *stream)
{
  return gather_strips(ctx, comm, packed_dev, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height,
                       strip_rows, root_in_place, 1, MRT_PLANES_ALL, stream, true);
}

int mrt_gather_strips_planes(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                             void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                             int32_t root_in_place, int32_t planes, void *stream)
{
  return gather_strips(ctx, comm, packed_dev, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height,
                       strip_rows, root_in_place, 1, planes, stream, true);
}

int mrt_gather_strips_root(mrt_ctx *ctx, mrt_comm *comm, const void *packed_dev, void *staging_dev, void *frame_rgb_dev,
                           void *frame_id_dev, void *frame_t_dev, int32_t width, int32_t height, int32_t strip_rows,
                           int32_t root_span, int32_t planes, void *stream)
{
  return gather_strips(ctx, comm, packed_dev, staging_dev, frame_rgb_dev, frame_id_dev, frame_t_dev, width, height,
                       strip_rows, 1, root_span, planes, stream, true);
}

int mrt_render_multi(mrt_ctx *const *ctxs, mrt_comm *const *comms, int32_t n, const mrt_camera *cam,
                     const mrt_render_params *params, float *rgb, int32_t *id, float *t, mrt_stats *stats)
{
  if (!ctxs || n < 1 || n > mrt::MAX_RANKS || !ctxs[0]) return MRT_ERR_INVALID;
  mrt_ctx *c0 = ctxs[0];
  if (!cam || !params) return MRT_ERR_INVALID;
  if (!rgb) return fail(c0, MRT_ERR_INVALID, "rgb buffer is required");
  for (int r = 0; r < n; ++r) {
    if (!ctxs[r]) return fail(c0, MRT_ERR_INVALID, "mrt_render_multi: null context");
    if (n > 1 && (!comms || !comms[r] || comms[r]->ctx != ctxs[r] || comms[r]->rank != r || comms[r]->nranks != n))
      return fail(c0, MRT_ERR_INVALID, "mrt_render_multi: comms must be what mrt_comm_create_local returned for ctxs");
  }
  const int W = params->width, H = params->height, strip_rows = params->part.strip_rows;
  uint64_t off[mrt::MAX_RANKS], len[mrt::MAX_RANKS];
  mrt::RenderArgs A[mrt::MAX_RANKS];
  // every rank's arguments first: nothing is enqueued before all of them are known to be valid
  for (int r = 0; r < n; ++r) {
    mrt_render_params p = *params;
    p.part = mrt_partition{r, n, strip_rows, (n > 1 && r == 0) ? 1 : 0};
    int rc = fill_args(ctxs[r], cam, &p, A[r]);
    if (rc != MRT_OK) { if (r) fail(c0, rc, ctxs[r]->err); return rc; }
  }
  // only the planes the caller asked for are rendered by the other ranks and travel (rgb alone: 12 of 20 bytes a pixel)
  const int32_t planes = MRT_PLANE_RGB | (id ? MRT_PLANE_ID : 0) | (t ? MRT_PLANE_T : 0);
  const int pd = plane_dwords(planes);
  const uint64_t staging_dwords = staging_layout(W, H, n, strip_rows, 0, pd, off, len);
  const size_t npix = (size_t)W * H;
  for (int r = 0; r < n; ++r) {
    mrt_ctx *ctx = ctxs[r];
    const size_t lpix = (size_t)A[r].local_rows * W;
    // rank 0: the frame planes, then the staging area; other ranks: their packed rgb | id | t planes
    const size_t need = (r == 0 ? (npix * 5 + (size_t)staging_dwords) : lpix * (size_t)pd) * 4 + 256;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->out_bytes < need) {
      HIP_TRY(ctx, drain(ctx));
      if (ctx->out_mem) { HIP_TRY(ctx, hipFree(ctx->out_mem)); ctx->out_mem = nullptr; ctx->out_bytes = 0; }
      HIP_TRY(ctx, hipMalloc(&ctx->out_mem, need));
      ctx->out_bytes = need;
    }
    const size_t ppix = r == 0 ? npix : lpix;
    A[r].rgb = (float *)ctx->out_mem;
    if (r == 0) { // the frame planes: all three have their place, whatever travels
      A[r].id = id ? (int32_t *)((char *)ctx->out_mem + ppix * 12) : nullptr;
      A[r].t = t ? (float *)((char *)ctx->out_mem + ppix * 16) : nullptr;
    } else {      // the packed buffer: rgb | id (if gathered) | t (if gathered)
      A[r].id = id ? (int32_t *)((char *)ctx->out_mem + ppix * 12) : nullptr;
      A[r].t = t ? (float *)((char *)ctx->out_mem + ppix * (id ? 16 : 12)) : nullptr;
    }
    A[r].stats = stats ? ctx->stats_dev : nullptr;
    A[r].stats_slots = STATS_SLOTS;
    HIP_TRY(ctx, order_after_last(ctx, ctx->stream));
    if (A[r].stats) HIP_TRY(ctx, hipMemsetAsync(A[r].stats, 0, STATS_SLOTS * 8 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    HIP_TRY(ctx, launch_by_mode(ctx, A[r], ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  }
  if (n > 1) {
    // The exchange of all ranks from this one thread: ONE group -- RCCL connects a pair of ranks when both sides have
    // posted, and both sides are this thread.  Sends and receives are enqueued when the group closes, so rank 0's
    // scatter is enqueued after that.
    const Rccl *R = rccl(nullptr);
    if (!R) return fail(c0, MRT_ERR_HIP, "RCCL is not open");
    RCCL_TRY(c0, R, R->GroupStart());
    int rc1 = MRT_OK;
    for (int r = 0; r < n && rc1 == MRT_OK; ++r) {
      mrt_ctx *ctx = ctxs[r];
      void *staging = r == 0 ? (char *)ctx->out_mem + npix * 20 : nullptr;
      rc1 = gather_strips(ctx, comms[r], r == 0 ? nullptr : ctx->out_mem, staging, r == 0 ? A[0].rgb : nullptr,
                          r == 0 ? A[0].id : nullptr, r == 0 ? A[0].t : nullptr, W, H, strip_rows, 1, 1, planes, ctx->stream, false);
      if (rc1 != MRT_OK && r) fail(c0, rc1, ctx->err);
    }
    ncclResult_t r2 = R->GroupEnd();
    if (rc1 != MRT_OK) return rc1;
    if (r2 != ncclSuccess) return rccl_fail(c0, R, r2, "ncclGroupEnd");
    if (staging_dwords) {
      rc1 = mrt_scatter_strips_planes(c0, (char *)c0->out_mem + npix * 20, A[0].rgb, A[0].id, A[0].t, W, H, n, strip_rows, 0,
                                      planes, c0->stream);
      if (rc1 != MRT_OK) return rc1;
    }
  }
  HIP_TRY(c0, hipSetDevice(c0->device));
  HIP_TRY(c0, hipMemcpyAsync(rgb, A[0].rgb, npix * 12, hipMemcpyDeviceToHost, c0->stream));
  if (id) HIP_TRY(c0, hipMemcpyAsync(id, A[0].id, npix * 4, hipMemcpyDeviceToHost, c0->stream));
  if (t) HIP_TRY(c0, hipMemcpyAsync(t, A[0].t, npix * 4, hipMemcpyDeviceToHost, c0->stream));
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float kernel_ms = 0.0f;
  std::vector<unsigned long long> slots;
  if (stats) {
    try { slots.resize((size_t)STATS_SLOTS * 8); } catch (...) { return fail(c0, MRT_ERR_INVALID, "out of host memory"); }
  }
  for (int r = n - 1; r >= 0; --r) { // rank 0 last: its stream also carries the frame copy
    mrt_ctx *ctx = ctxs[r];
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (stats) HIP_TRY(ctx, hipMemcpyAsync(slots.data(), ctx->stats_dev, slots.size() * sizeof(unsigned long long),
                                           hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (stats) {
      for (int s = 0; s < STATS_SLOTS; ++s)
        for (int k = 0; k < 8; ++k) h[k] += slots[(size_t)s * 8 + k];
      float ms = 0.0f;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      kernel_ms = ms > kernel_ms ? ms : kernel_ms;
    }
  }
  if (stats) {
    stats->primary_rays = h[0]; stats->shadow_rays = h[1]; stats->hit_pixels = h[2]; stats->occluded = h[3];
    stats->exact_tests = h[4]; stats->clusters_opened = h[5];
    stats->kernel_ms = kernel_ms;
  }
  return MRT_OK;
}

} // extern "C"
