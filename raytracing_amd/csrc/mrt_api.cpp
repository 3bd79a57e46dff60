// mrt_api.cpp -- C-ABI of libmrt.so (declared in include/mrt.h).
//
// Thin layer: owns one GPU context (device, stream, scene SoA buffers in HBM,
// output staging buffers), validates shapes on the host before any launch, and
// turns HIP failures into status codes + an error string.  It never exits and
// never throws; the print-and-exit convention of the reference's
// src/check_error.cu:7-15 is kept by the C++ host layer (checkRtErrors).

#include "mrt.h"
#include "mrt_internal.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
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

  // u = col / W, v = row / H tables of the small-scene kernel, for the frame size of the last render
  float *uv_mem = nullptr;
  int uv_w = 0, uv_h = 0, uv_nu = 0;
  // host copy of the triangles of a small scene (16 floats each, device order): the small-scene kernel's
  // launcher derives the camera-dependent culling functions from it for every render
  std::vector<float> small_tri16, small_sph;
  std::vector<int32_t> small_tri_obj, small_sph_obj;

  // per-pixel XORWOW streams of the diffuse bounce (mrt_rng.hip): jump matrices, and the xorshift words of
  // streams 0..rng_n-1 for seed rng_seed (5 streams of rng_n words), built on first need
  uint32_t *rng_jump = nullptr;
  uint32_t *rng_mem = nullptr;
  uint32_t rng_n = 0;
  uint64_t rng_seed = 1984, rng_built_seed = 0; // cuda_utils/curand_state.cuh:22

};

// the blocking entry point spreads its ray counters over this many copies (same-address atomics serialise: with
// one copy the default frame's 16 200 waves spent 0.7 ms on them) and adds them up on the host
enum { STATS_SLOTS = 256 };

static thread_local std::string g_create_err;

static int fail(mrt_ctx *ctx, int code, const std::string &msg)
{
  if (ctx) ctx->err = msg; else g_create_err = msg;
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
// reference loops), 1 automatic (default), 2 one-level tile culling, 3 / 4 two-level culling with 4 / 8 waves,
// 5 the small-scene kernel (at most 64 primitives; larger scenes fall back to 2).
// All of them produce identical buffers (tests compare them).
static int use_cull()
{
  static const int v = [] {
    const char *e = getenv("MRT_CULL");
    const int x = e ? atoi(e) : 1;
    return (x < 0 || x > 5) ? 1 : x;
  }();
  return v;
}

// The small-scene kernel reads u = col / W and v = row / H from tables; they are rebuilt (on the stream of the
// render, followed by one synchronisation so that later renders on other streams may use them) only when the
// frame size changes.
static hipError_t ensure_uv_tables(mrt_ctx *ctx, int W, int H, hipStream_t st)
{
  if (ctx->uv_mem && ctx->uv_w == W && ctx->uv_h == H) return hipSuccess;
  hipError_t e;
  if (ctx->uv_mem) {
    if ((e = hipDeviceSynchronize()) != hipSuccess) return e; // a render in flight may still read the old tables
    (void)hipFree(ctx->uv_mem);
    ctx->uv_mem = nullptr;
  }
  const int nu = (W + 31) / 32 * 32; // whole tiles of up to 32 pixels
  if ((e = hipMalloc((void **)&ctx->uv_mem, sizeof(float) * ((size_t)nu + H))) != hipSuccess) return e;
  if ((e = mrt::launch_uv_table(ctx->uv_mem, nu, W, ctx->uv_mem + nu, H, H, st)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
  ctx->uv_w = W; ctx->uv_h = H; ctx->uv_nu = nu;
  return hipSuccess;
}

// Streams 0..npix-1 (the stream of a pixel is its index in the FULL frame, src/main.cu:133, whatever strip
// partition renders it).  Built on the context's stream and synchronised once, like the u/v tables.
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
    if (ctx->rng_mem) {
      if ((e = hipDeviceSynchronize()) != hipSuccess) return e; // a render in flight may still read the old buffer
      (void)hipFree(ctx->rng_mem);
      ctx->rng_mem = nullptr; ctx->rng_n = 0;
    }
    const size_t n = (npix + 1023) / 1024 * 1024;
    if ((e = hipMalloc((void **)&ctx->rng_mem, 5 * n * sizeof(uint32_t))) != hipSuccess) return e;
    ctx->rng_n = (uint32_t)n;
  } else if ((e = hipDeviceSynchronize()) != hipSuccess) return e; // new seed: same buffer, rebuilt
  if ((e = mrt::launch_rng_init(ctx->rng_jump, ctx->rng_mem, ctx->rng_n, ctx->rng_n, ctx->rng_seed, ctx->stream)) != hipSuccess) return e;
  if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return e;
  ctx->rng_built_seed = ctx->rng_seed;
  return hipSuccess;
}

static hipError_t launch_by_mode(mrt_ctx *ctx, mrt::RenderArgs &A, hipStream_t st)
{
  if (A.max_depth > 0) { // bounce chains: one kernel, every ray x every primitive
    if (ctx->any_diffuse) {
      hipError_t e = ensure_rng_states(ctx, (size_t)A.W * A.H);
      if (e != hipSuccess) return e;
      uint32_t v[5];
      mrt::rng_seed_state(ctx->rng_seed, v, &A.rng_d);
      A.rng = ctx->rng_mem; A.rng_stride = ctx->rng_n;
    }
    // MRT_OPT_CULL = 0 keeps the small-scene culling data out of it (every ray walks the hierarchy: cross-check)
    const bool small = (long)A.sc.ns + A.sc.nt <= 64 && ctx->cull != 0;
    mrt::SmallHost hs;
    if (small) {
      hs.ns = (int)ctx->small_sph_obj.size(); hs.nt = (int)ctx->small_tri_obj.size();
      hs.sph = ctx->small_sph.data(); hs.sph_obj = ctx->small_sph_obj.data();
      hs.tri16 = ctx->small_tri16.data(); hs.tri_obj = ctx->small_tri_obj.data();
    }
    return mrt::launch_render_bounce(A, small ? &hs : nullptr, st);
  }
  int mode = ctx->cull;
  const bool small = (long)A.sc.ns + A.sc.nt <= 64;
  // measured: the small-scene kernel wins where it applies; with the primitive-side hierarchy the
  // one-level tile kernel wins everywhere else
  if (mode == 1) mode = small ? 5 : 2;
  if (mode == 5 && !small) mode = 2;
  switch (mode) {
  case 0: return mrt::launch_render(A, st);
  case 2: return mrt::launch_render_cull(A, st);
  case 3: return mrt::launch_render_cull2(A, 4, st);
  case 4: return mrt::launch_render_cull2(A, 8, st);
  default: {
    hipError_t e = ensure_uv_tables(ctx, A.W, A.H, st);
    if (e != hipSuccess) return e;
    A.utab = ctx->uv_mem; A.vtab = ctx->uv_mem + ctx->uv_nu;
    mrt::SmallHost hs;
    hs.ns = (int)ctx->small_sph_obj.size(); hs.nt = (int)ctx->small_tri_obj.size();
    hs.sph = ctx->small_sph.data(); hs.sph_obj = ctx->small_sph_obj.data();
    hs.tri16 = ctx->small_tri16.data(); hs.tri_obj = ctx->small_tri_obj.data();
    return mrt::launch_render_small(A, hs, st);
  }
  }
}

// ---- primitive-side hierarchy (camera independent, built once per upload) ----
// Primitives of each type are stored in kd order (default; Morton order for ablation) of their bound centres, and every run of 64 consecutive
// primitives ("cluster") gets a bounding ball.  The culling kernels test clusters first.  Reordering is
// invisible in the results: hits are reported by object number, and the closest-hit rule
// (t, object number) lexicographic minimum does not depend on the visiting order.
static inline uint32_t spread10(uint32_t v)
{
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000ff;
  v = (v | (v << 8)) & 0x0300f00f;
  v = (v | (v << 4)) & 0x030c30c3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

// order[i] = index of the primitive stored at position i; bound(i, out4) gives centre + radius
template <class BoundFn> static void morton_order(size_t n, BoundFn bound, std::vector<uint32_t> &order)
{
  order.resize(n);
  for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
  if (n <= 64) return; // one cluster: order is irrelevant
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  std::vector<float> c(4 * n);
  for (size_t i = 0; i < n; ++i) {
    bound(i, &c[4 * i]);
    if (!std::isfinite(c[4 * i + 3])) continue;
    for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], c[4 * i + a]); hi[a] = std::fmax(hi[a], c[4 * i + a]); }
  }
  const double span = std::fmax(hi[0] - lo[0], std::fmax(hi[1] - lo[1], hi[2] - lo[2]));
  std::vector<uint64_t> key(n);
  for (size_t i = 0; i < n; ++i) {
    const float *b = &c[4 * i];
    if (!std::isfinite(b[3]) || !std::isfinite(b[0]) || !std::isfinite(b[1]) || !std::isfinite(b[2])) {
      key[i] = (uint64_t)1 << 40; // unbounded primitives go last, into clusters of their own kind
      continue;
    }
    // one scale for the three axes (the largest extent): a thin axis must not contribute high Morton bits
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
      const double f = span > 0 ? (b[a] - lo[a]) / span : 0.0;
      q[a] = (uint32_t)std::fmin(1023.0, std::fmax(0.0, f * 1023.0));
    }
    key[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  }
  std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
}

// Alternative order: recursive median split along the longest axis of the centres, with every split position a
// multiple of 64 (of 4096 while the range is larger than that), so that each run of 64 consecutive primitives
// is a leaf of a kd-tree and each run of 64 clusters a subtree: compact clusters, where a run of 64 along a
// Morton curve can straddle a long jump of the curve.  Unbounded primitives go last, as in morton_order.
static void kd_split(uint32_t *idx, size_t lo, size_t hi, const float *c)
{
  const size_t n = hi - lo;
  if (n <= 64) return;
  const size_t unit = n > 4096 ? 4096 : 64;
  size_t mid = lo + ((n / 2 + unit / 2) / unit) * unit;
  if (mid <= lo) mid = lo + unit;
  if (mid >= hi) mid = lo + ((n - 1) / unit) * unit;
  if (mid <= lo || mid >= hi) return;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = lo; i < hi; ++i)
    for (int a = 0; a < 3; ++a) { const float v = c[4 * idx[i] + a]; mn[a] = std::fmin(mn[a], v); mx[a] = std::fmax(mx[a], v); }
  int axis = 0;
  if (mx[1] - mn[1] > mx[axis] - mn[axis]) axis = 1;
  if (mx[2] - mn[2] > mx[axis] - mn[axis]) axis = 2;
  std::nth_element(idx + lo, idx + mid, idx + hi,
                   [&](uint32_t a, uint32_t b) { return c[4 * a + axis] < c[4 * b + axis]; });
  kd_split(idx, lo, mid, c);
  kd_split(idx, mid, hi, c);
}

template <class BoundFn> static void kd_order(size_t n, BoundFn bound, std::vector<uint32_t> &order)
{
  order.clear();
  order.reserve(n);
  std::vector<float> c(4 * n);
  std::vector<uint32_t> unbounded;
  for (size_t i = 0; i < n; ++i) {
    bound(i, &c[4 * i]);
    const float *b = &c[4 * i];
    if (std::isfinite(b[3]) && std::isfinite(b[0]) && std::isfinite(b[1]) && std::isfinite(b[2])) order.push_back((uint32_t)i);
    else unbounded.push_back((uint32_t)i);
  }
  if (n > 64) kd_split(order.data(), 0, order.size(), c.data());
  order.insert(order.end(), unbounded.begin(), unbounded.end());
}

// MRT_ORDER=morton selects the Morton order (ablation); default: kd order
template <class BoundFn> static void primitive_order(size_t n, BoundFn bound, std::vector<uint32_t> &order)
{
  static const bool morton = [] { const char *e = getenv("MRT_ORDER"); return e && std::strcmp(e, "morton") == 0; }();
  if (morton) morton_order(n, bound, order);
  else kd_order(n, bound, order);
}

// bounding ball of each run of 64 balls (streams x,y,z,r of n floats) -> 4 streams of ceil(n/64) floats
static void cluster_bounds(const float *x, const float *y, const float *z, const float *r, size_t n, float *out)
{
  const size_t nc = (n + 63) / 64;
  for (size_t c = 0; c < nc; ++c) {
    const size_t b = c * 64, e = std::min(n, b + 64);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool finite = true;
    for (size_t i = b; i < e; ++i) {
      const double p[3] = {x[i], y[i], z[i]}, rr = r[i];
      if (!std::isfinite(rr) || !std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) { finite = false; break; }
      for (int a = 0; a < 3; ++a) { lo[a] = std::fmin(lo[a], p[a] - rr); hi[a] = std::fmax(hi[a], p[a] + rr); }
    }
    float o[4] = {0.0f, 0.0f, 0.0f, INFINITY};
    if (finite) {
      const double m[3] = {0.5 * (lo[0] + hi[0]), 0.5 * (lo[1] + hi[1]), 0.5 * (lo[2] + hi[2])};
      const float mf[3] = {(float)m[0], (float)m[1], (float)m[2]};
      double R = 0.0;
      for (size_t i = b; i < e; ++i) {
        const double d[3] = {x[i] - (double)mf[0], y[i] - (double)mf[1], z[i] - (double)mf[2]};
        R = std::fmax(R, std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + (double)r[i]);
      }
      R *= 1.000001;
      if (std::isfinite(R) && R < 1e30) { o[0] = mf[0]; o[1] = mf[1]; o[2] = mf[2]; o[3] = std::nextafter((float)R, INFINITY); }
    }
    for (int k = 0; k < 4; ++k) out[k * nc + c] = o[k];
  }
}

extern "C" {

int mrt_ctx_set_option(mrt_ctx *ctx, int option, int value)
{
  if (!ctx) return MRT_ERR_INVALID;
  if (option == MRT_OPT_CULL) {
    if (value < 0 || value > 5) return fail(ctx, MRT_ERR_INVALID, "MRT_OPT_CULL takes 0..5");
    ctx->cull = value;
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
  auto bail = [&](hipError_t err, const char *what) {
    int rc = hip_fail(nullptr, err, what);
    mrt_ctx_destroy(ctx);
    return rc;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
  if ((e = hipMalloc((void **)&ctx->stats_dev, STATS_SLOTS * 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
  *out = ctx;
  return MRT_OK;
}

int mrt_ctx_destroy(mrt_ctx *ctx)
{
  if (!ctx) return MRT_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->scene_mem) (void)hipFree(ctx->scene_mem);
  if (ctx->uv_mem) (void)hipFree(ctx->uv_mem);
  if (ctx->rng_mem) (void)hipFree(ctx->rng_mem);
  if (ctx->rng_jump) (void)hipFree(ctx->rng_jump);
  if (ctx->out_mem) (void)hipFree(ctx->out_mem);
  if (ctx->stats_dev) (void)hipFree(ctx->stats_dev);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MRT_OK;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Conservative bounding sphere of the region Triangle::hit accepts (core/triangle.cuh:89-91): the set of
// points p with n_i.p - k_i > -1e-6 for the three edge planes, near the plane pn.p = kp.  Derived from the
// 16 precomputed floats only (the hit test never sees vertices, and for degenerate triangles -- e.g. a
// plane through the world origin, SURVEY section 0.6 -- the accepted region is NOT the geometric triangle).
// Unbounded, empty or ill-conditioned regions get radius +inf, i.e. are never culled.
static void tri_bound(const float q[16], float out[4])
{
  typedef long double R; // x87 extended on the host: the sign decisions below need more than double at det ~ 1e-9
  out[0] = out[1] = out[2] = 0.0f;
  out[3] = INFINITY;
  R pn[3] = {q[0], q[1], q[2]}, kp = q[3], n[3][3], k[3];
  for (int i = 0; i < 3; ++i) { n[i][0] = q[4 + 4 * i]; n[i][1] = q[5 + 4 * i]; n[i][2] = q[6 + 4 * i]; k[i] = q[7 + 4 * i]; }
  auto norm = [](const R v[3]) { return sqrtl(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
  auto dot = [](const R a[3], const R b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  for (int i = 0; i < 16; ++i) if (!std::isfinite(q[i])) return;
  const R pl = norm(pn);
  if (!(pl > 1e-30)) return;
  R nl[3];
  for (int i = 0; i < 3; ++i) { nl[i] = norm(n[i]); if (!(nl[i] > 1e-30)) return; }
  // x with pn.x = a0, n_i.x = a1, n_j.x = a2 (Cramer); false if the three planes are near-dependent.  *rel is
  // the determinant over the product of the row norms: x carries a relative error of a few 1e-19 / *rel
  auto solve = [&](int i, int j, R a0, R a1, R a2, R x[3], R* rel) {
    const R *r0 = pn, *r1 = n[i], *r2 = n[j];
    const R c0[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const R c1[3] = {r2[1] * r0[2] - r2[2] * r0[1], r2[2] * r0[0] - r2[0] * r0[2], r2[0] * r0[1] - r2[1] * r0[0]};
    const R c2[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const R det = dot(r0, c0);
    if (!(fabsl(det) > 1e-9 * pl * nl[i] * nl[j])) return false;
    *rel = fabsl(det) / (pl * nl[i] * nl[j]);
    for (int a = 0; a < 3; ++a) x[a] = (a0 * c0[a] + a1 * c1[a] + a2 * c2[a]) / det;
    return std::isfinite(x[0]) && std::isfinite(x[1]) && std::isfinite(x[2]);
  };
  const int pi[3] = {0, 1, 2}, pj[3] = {1, 2, 0}, pm[3] = {2, 0, 1};
  R V[3][3], vrel[3], pmax = 0.0;
  for (int e = 0; e < 3; ++e) {
    if (!solve(pi[e], pj[e], kp, k[pi[e]], k[pj[e]], V[e], &vrel[e])) return;
    pmax = fmaxl(pmax, norm(V[e]));
  }
  // The three half-planes bound a triangle <=> every vertex lies strictly inside the third one.  The sign is
  // what matters: with a vertex even 1e-8 (relative) on the wrong side, the region is an unbounded wedge -- the
  // float edge planes of a sliver do that, and the hit test then accepts points tens of units from the geometric
  // triangle.  Anything not clearly inside (by ~40x the error of the solve, and at least 1e-12 of the
  // magnitudes) is "unbounded".  (Edge planes are dir x point_inside with a POSITION vector, so all three contain
  // the direction of that position and the solve degrades as the triangle's plane nears the world origin: a
  // terrain around y = 0 has determinants down to 1e-8, which is why the floor is tied to *rel and not fixed.)
  for (int e = 0; e < 3; ++e) {
    const int m = pm[e];
    const R tol = fmaxl(1e-12L, 2e-17L / vrel[e]);
    if (!(dot(n[m], V[e]) - k[m] >= tol * (nl[m] * pmax + fabsl(k[m])))) return;
  }
  // push every edge plane outwards by the test's epsilon plus a bound on its fp32 evaluation error at points of
  // norm <= pb; a thin triangle grows a lot under that (by the push over the sine of its sharpest angle), so pb
  // is raised until the pushed-out vertices lie within it
  R pb = 1.5 * pmax + 1.0;
  R W[3][3], c[3] = {0, 0, 0};
  bool settled = false;
  for (int round = 0; round < 6 && !settled; ++round) {
    R kd[3], wmax = 0.0;
    for (int i = 0; i < 3; ++i) kd[i] = k[i] - (1e-6 + 2e-6 * (nl[i] * pb + fabsl(k[i])));
    for (int e = 0; e < 3; ++e) {
      R rel;
      if (!solve(pi[e], pj[e], kp, kd[pi[e]], kd[pj[e]], W[e], &rel)) return;
      wmax = fmaxl(wmax, norm(W[e]));
    }
    if (wmax <= pb) settled = true;
    else pb = 1.5 * wmax + 1.0;
  }
  if (!settled) return;
  for (int e = 0; e < 3; ++e)
    for (int a = 0; a < 3; ++a) c[a] += W[e][a] / 3.0;
  R rad = 0.0;
  for (int e = 0; e < 3; ++e) {
    const R d[3] = {W[e][0] - c[0], W[e][1] - c[1], W[e][2] - c[2]};
    rad = fmaxl(rad, norm(d));
  }
  // hit points are float-rounded and only approximately on the plane
  rad = (rad + 1e-5 * (pb + fabsl(kp) / pl) + 1e-6 * norm(c)) * 1.0001;
  if (!std::isfinite(rad) || !(rad < 1e30)) return;
  out[0] = (float)c[0]; out[1] = (float)c[1]; out[2] = (float)c[2];
  out[3] = std::nextafter((float)rad, INFINITY);
}

// ---- shadow-ray edge functions (small scenes; scene and lights only, built once per upload) ----
// Shadow ray of light l from hit point p, p' the point where it meets the triangle's plane (n, k):
// e_i(p') rdn = H_i(p), affine in p, with rdn = n.sd (directional light, sd = -direction) or n.(L - p) (point
// light at L); for a ray the exact test accepts, sign(rdn) is fixed: sign(n.sd), resp. sign(n.L - k).
// Stored per edge: g = s grad H_i, off = s H_i(0), gn >= |g|_1, and margin coefficients a, b; then
//    g.m + off + rho gn + margin < 0   =>   no shadow ray from the ball (m, rho) of hit points passes edge i,
// margin = a + b Pmax (directional), (|L|_1 + Pmax)(a + b Pmax) (point), Pmax >= |hit point|_1.  The margins
// are 1e-4 of the magnitudes entering the exact test (which rounds at 6e-8) plus its own -1e-6 tolerance.
// Slots 21, 22: point lights only cull while |n.L - k| > slot21' + slot22 Pmax (a light in the plane breaks
// the sign argument); stored as |n.L - k| - thr0 and thr1.
enum { SH_STRIDE = 23, SMALL_MAX_SHADOW_LIGHTS = 4 };

static void shadow_functions(const float q[16], int kind, const float lv[3], float out[SH_STRIDE])
{
  auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto n1f = [](const double *a) { return std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]); };
  auto up = [](double x) { return (float)(x * (1.0 + 1e-6)); }; // magnitudes: round up
  const double n[3] = {q[0], q[1], q[2]}, kp = q[3], nn1 = n1f(n);
  const double L[3] = {lv[0], lv[1], lv[2]};
  double sg, rdn = 0, nlk = 0, S0 = 0;
  bool always = false;
  if (kind == MRT_LIGHT_DIRECTIONAL) {
    // the sign the kernel sees: same fp32 expression as the exact test's denominator
    const float sdx = -lv[0], sdy = -lv[1], sdz = -lv[2];
    const float rdnf = sdx * q[0] + (sdy * q[1] + sdz * q[2]);
    sg = rdnf >= 0.0f ? 1.0 : -1.0;
    always = (rdnf == 0.0f); // the exact test rejects rdn == 0
    const double sd[3] = {-L[0], -L[1], -L[2]};
    rdn = dot(sd, n);
    S0 = n1f(sd);
    out[21] = INFINITY;
    out[22] = 0.0f;
  } else {
    nlk = dot(n, L) - kp;
    sg = nlk >= 0 ? 1.0 : -1.0;
    out[21] = (float)((std::fabs(nlk) - 1e-5 * (std::fabs(kp) + nn1 * n1f(L))) * (1.0 - 1e-6));
    out[22] = up(1e-5 * nn1);
  }
  for (int i = 0; i < 3; ++i) {
    const double m[3] = {q[4 + 4 * i], q[5 + 4 * i], q[6 + 4 * i]}, c = q[7 + 4 * i], m1 = n1f(m);
    double g[3], off, a, b;
    if (kind == MRT_LIGHT_DIRECTIONAL) {
      const double sd[3] = {-L[0], -L[1], -L[2]};
      const double mu = dot(m, sd), R0 = std::fabs(rdn);
      for (int x = 0; x < 3; ++x) g[x] = sg * (rdn * m[x] - mu * n[x]);
      off = sg * (kp * mu - c * rdn);
      a = 1e-6 * R0 + 1e-4 * (std::fabs(c) * R0 + m1 * S0 * std::fabs(kp));
      b = 1e-4 * m1 * (R0 + S0 * nn1);
    } else {
      const double mL = dot(m, L), nL = dot(n, L), cm = c - mL;
      for (int x = 0; x < 3; ++x) g[x] = sg * (nlk * m[x] + cm * n[x]);
      off = sg * (kp * mL - c * nL);
      a = 1e-6 * nn1 + 1e-4 * (std::fabs(c) * nn1 + m1 * std::fabs(kp));
      b = 2e-4 * m1 * nn1;
    }
    if (always) { g[0] = g[1] = g[2] = 0; off = -3.0e38; a = 0; b = 0; }
    out[7 * i + 0] = (float)g[0]; out[7 * i + 1] = (float)g[1]; out[7 * i + 2] = (float)g[2];
    out[7 * i + 3] = (float)off;
    out[7 * i + 4] = up(n1f(g));
    out[7 * i + 5] = up(a); out[7 * i + 6] = up(b);
  }
}

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
      ctx->small_sph.clear(); ctx->small_sph_obj.clear();
      if (ns + nt <= 64)
        for (size_t i = 0; i < ns; ++i) {
          ctx->small_sph.insert(ctx->small_sph.end(), {cx[i], cy[i], cz[i], r[i]});
          ctx->small_sph_obj.push_back(so[i]);
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
    primitive_order(nt, [&](size_t i, float *b) { std::memcpy(b, &tb[4 * i], 16); }, order);
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
      if (small) { ctx->small_tri16.assign(rec, rec + 16 * nt); ctx->small_tri_obj.assign(to, to + nt); }
      else { ctx->small_tri16.clear(); ctx->small_tri_obj.clear(); }
    }
    catch (...) { return fail(ctx, MRT_ERR_INVALID, "out of host memory"); }
    float *ct = (float *)(img.data() + o_ct);
    cluster_bounds(bnd, bnd + nt, bnd + 2 * nt, bnd + 3 * nt, nt, ct);
    cluster_bounds(ct, ct + nct, ct + 2 * nct, ct + 3 * nct, nct, (float *)(img.data() + o_sct));
  }
  put(o_mk, s->mat_kind, no);      put(o_mrgb, s->mat_rgb, no * 12);
  put(o_lk, s->light_kind, nl);    put(o_lv, s->light_vec, nl * 12); put(o_lc, s->light_rgb, nl * 12);

  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->scene_bytes < total) {
    if (ctx->scene_mem) { HIP_TRY(ctx, hipFree(ctx->scene_mem)); ctx->scene_mem = nullptr; ctx->scene_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->scene_mem, total));
    ctx->scene_bytes = total;
  }
  ctx->have_scene = false;
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
  d.sph_sclu = (const float *)(b + o_scs); d.nscs = (int32_t)nscs;
  d.tri_sclu = (const float *)(b + o_sct); d.nsct = (int32_t)nsct;
  d.mat_kind = b + o_mk; d.mat_rgb = (const float *)(b + o_mrgb); d.nobj = (int32_t)no;
  d.light_kind = b + o_lk; d.light_vec = (const float *)(b + o_lv); d.light_rgb = (const float *)(b + o_lc);
  d.nl = (int32_t)nl;
  ctx->any_diffuse = false;
  for (size_t i = 0; i < no; ++i) ctx->any_diffuse |= (s->mat_kind[i] == MRT_MAT_DIFFUSE);
  ctx->have_scene = true;
  return MRT_OK;
}

static bool partition_ok(const mrt_partition &p)
{
  return p.num_ranks >= 1 && p.rank >= 0 && p.rank < p.num_ranks && p.strip_rows >= 8 && (p.strip_rows % 8) == 0;
}

static int local_rows_of(int H, const mrt_partition &p)
{
  const int nstrips = (H + p.strip_rows - 1) / p.strip_rows;
  int rows = 0;
  for (int s = p.rank; s < nstrips; s += p.num_ranks) {
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
  A.rank = p->part.rank; A.nranks = p->part.num_ranks; A.strip_rows = p->part.strip_rows;
  A.local_rows = local_rows_of(p->height, p->part);
  A.spp = p->spp;
  A.max_depth = p->max_depth;
  A.stats_slots = 1;
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
  if (stats_dev) { // counted in the context's spread copies, then added up into the caller's eight words
    A.stats = ctx->stats_dev;
    A.stats_slots = STATS_SLOTS;
    HIP_TRY(ctx, hipMemsetAsync(A.stats, 0, STATS_SLOTS * 8 * sizeof(unsigned long long), st));
  }
  HIP_TRY(ctx, launch_by_mode(ctx, A, st));
  if (stats_dev) HIP_TRY(ctx, mrt::launch_stats_reduce(ctx->stats_dev, STATS_SLOTS, (unsigned long long *)stats_dev, st));
  return MRT_OK;
}

int mrt_render(mrt_ctx *ctx, const mrt_camera *cam, const mrt_render_params *params, float *rgb, int32_t *id, float *t,
               mrt_stats *stats)
{
  mrt::RenderArgs A;
  int rc = fill_args(ctx, cam, params, A);
  if (rc != MRT_OK) return rc;
  if (!rgb) return fail(ctx, MRT_ERR_INVALID, "rgb buffer is required");
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

int mrt_assemble_strips(mrt_ctx *ctx, const void *gathered_dev, size_t rank_stride_elems, void *frame_dev, int32_t width,
                        int32_t height, int32_t elem_dwords, int32_t num_ranks, int32_t strip_rows, void *stream)
{
  if (!ctx || !gathered_dev || !frame_dev) return MRT_ERR_INVALID;
  if (width <= 0 || height <= 0 || elem_dwords <= 0 || num_ranks <= 0 || strip_rows <= 0)
    return fail(ctx, MRT_ERR_INVALID, "bad assemble shape");
  mrt_partition p0{0, num_ranks, strip_rows};
  if (!partition_ok(p0)) return fail(ctx, MRT_ERR_INVALID, "bad strip partition");
  // rank 0 holds the most rows; every rank's slab must fit in the stride
  if ((size_t)local_rows_of(height, p0) * width * elem_dwords > rank_stride_elems)
    return fail(ctx, MRT_ERR_INVALID, "rank stride smaller than a rank's strip buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)stream;
  HIP_TRY(ctx, mrt::launch_assemble((const uint32_t *)gathered_dev, rank_stride_elems, (uint32_t *)frame_dev, width,
                                    height, elem_dwords, num_ranks, strip_rows, st));
  return MRT_OK;
}

int mrt_intersect(mrt_ctx *ctx, int prim_kind, const float *prim, const float *rays, float tmin, float tmax, int64_t n,
                  uint8_t *hit, float *t, float *point, float *normal)
{
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
    if ((e = mrt::launch_intersect(prim_kind, d_prim, d_rays, tmin, tmax, n, d_hit, d_t, d_point, d_normal,
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
