/*
 * rt_oracle.c -- CPU oracle for the render hot path.  TEST INFRASTRUCTURE ONLY
 * (see rt_oracle.h for who may use it and how it is pinned).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * Every arithmetic helper below fixes ONE evaluation order; the HIP kernels and
 * the C++ host layer use the same order, which is what makes hit/ID buffers
 * bit-comparable.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vec3 */

typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_from(const float p[3]) { return v3_make(p[0], p[1], p[2]); }
static inline void v3_to(v3 a, float p[3]) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
/* Eigen fixed-size redux of length 3: first half (1 element) + second half (2 elements) */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline v3 v3_cross(v3 a, v3 b)
{
  return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 v3_normalized(v3 a)
{
  float z = v3_dot(a, a);
  if (z > 0.0f) return v3_div(a, sqrtf(z));
  return a;
}

/* ------------------------------------------------------------------ core types */

typedef struct { v3 o, d; } ray_t;                      /* core/ray.cuh:7-30 */
typedef struct { float t; v3 point, normal; } hit_t;   /* core/hit_result.cuh:6-10 */
/* core/interval.cuh:4-48; lo_open / hi_open = 0 is the closed / closed mode the render path uses */
typedef struct { float lo, hi; int lo_open, hi_open; } interval_t;

/* core/interval.cuh:25-42: the four modes */
static inline int interval_contains(interval_t i, float f)
{
  const int lo = i.lo_open ? (i.lo < f) : (i.lo <= f);
  const int hi = i.hi_open ? (f < i.hi) : (f <= i.hi);
  return lo && hi;
}
/* core/ray.cuh:22-25 */
static inline v3 ray_at(const ray_t *r, float t) { return v3_add(r->o, v3_scale(t, r->d)); }

typedef struct { v3 c; float r; } sphere_t;             /* core/sphere.cuh:116-117 */
typedef struct { v3 n; float k; } disc_t;               /* core/triangle.cuh:50-51 */
typedef struct {                                        /* core/triangle.cuh:118-129 */
  v3 pn; float kp;
  disc_t ab_c, bc_a, ca_b;
  v3 a, b, c; /* kept so the AoS object has the reference's footprint; unused by hit */
} triangle_t;
typedef struct { float k; v3 n; } plane_t;              /* core/plane.cuh:37-38 */

typedef struct {                                        /* core/scene_object.cuh:18-93 */
  int geom_kind;
  union { sphere_t s; plane_t p; triangle_t t; } g;
  int mat_kind;
  v3 color;
} object_t;

typedef struct { int kind; v3 v; v3 color; } light_t;   /* v = dir_ (normalised) or pos_ */

struct orc_scene {
  object_t *obj; int nobj, cap_obj;
  light_t *light; int nlight, cap_light;
  uint64_t rng_seed; /* cuda_utils/curand_state.cuh:22: 1984 */
};

/* ------------------------------------------------------------------ geometry */

/* core/sphere.cuh:32-73 with compute_quadratic_params :89-114 */
static int sphere_hit(const sphere_t *s, const ray_t *r, interval_t I, hit_t *out)
{
  v3 oc = v3_sub(r->o, s->c);
  float a = v3_dot(r->d, r->d);
  float b = v3_dot(oc, r->d);
  float c = v3_dot(oc, oc) - s->r * s->r;
  float disc = b * b - a * c;

  if (disc > 0.0f) {
    float t = (-b - sqrtf(disc)) / a;
    if (interval_contains(I, t)) {
      out->point = ray_at(r, t);
      out->normal = v3_normalized(v3_sub(out->point, s->c));
      out->t = t;
      return 1;
    }
    t = (-b + sqrtf(disc)) / a;
    if (interval_contains(I, t)) {
      out->point = ray_at(r, t);
      out->normal = v3_normalized(v3_sub(out->point, s->c));
      out->t = t;
      return 1;
    }
  } else if (disc == 0.0f) {
    float t = (-b) / a;
    if (interval_contains(I, t)) {
      out->point = ray_at(r, t);
      out->normal = v3_normalized(v3_sub(out->point, s->c));
      out->t = t;
      return 1;
    }
  }
  return 0;
}

/* core/triangle.cuh:17-26 with compute_normal :35-41 and compute_intercept :44-48.
 * NB the edge-plane normal is dir x point_inside (a position vector), as the reference has it. */
static disc_t disc_make(v3 line_start, v3 line_end, v3 point_inside)
{
  disc_t d;
  v3 dir = v3_normalized(v3_sub(line_end, line_start));
  d.n = v3_cross(dir, point_inside);
  d.k = v3_dot(d.n, line_start);
  float side = (v3_dot(d.n, point_inside) - d.k) > -1e-6f ? 1.0f : -1.0f;
  d.n = v3_make(d.n.x * side, d.n.y * side, d.n.z * side);
  d.k = d.k * side;
  return d;
}

/* core/triangle.cuh:28-31 */
static inline int disc_inside(const disc_t *d, v3 p) { return (v3_dot(d->n, p) - d->k) > -1e-6f; }

/* core/triangle.cuh:58-66 */
static triangle_t triangle_make(v3 a, v3 b, v3 c)
{
  triangle_t t;
  memset(&t, 0, sizeof t);
  t.pn = v3_normalized(v3_cross(v3_sub(a, b), v3_sub(c, a)));
  t.ab_c = disc_make(a, b, c);
  t.bc_a = disc_make(b, c, a);
  t.ca_b = disc_make(c, a, b);
  t.kp = v3_dot(t.pn, a);
  t.a = a; t.b = b; t.c = c;
  return t;
}

/* core/triangle.cuh:77-103 */
static int triangle_hit(const triangle_t *tr, const ray_t *r, interval_t I, hit_t *out)
{
  float rdn = v3_dot(r->d, tr->pn);
  if (rdn != 0.0f) {
    float t = (tr->kp - v3_dot(r->o, tr->pn)) / rdn;
    v3 point = ray_at(r, t);
    int inside = disc_inside(&tr->ab_c, point) && disc_inside(&tr->bc_a, point) && disc_inside(&tr->ca_b, point);
    if (inside) {
      if (interval_contains(I, t)) {
        out->t = t;
        out->point = point;
        out->normal = (rdn < 0) ? tr->pn : v3_neg(tr->pn);
        return 1;
      }
    }
  }
  return 0;
}

/* core/scene_object.cuh:35-43 -> container/variant.cuh:234-262 (if-chain on the tag) */
static int object_hit(const object_t *o, const ray_t *r, interval_t I, hit_t *out)
{
  switch (o->geom_kind) {
  case ORC_GEOM_SPHERE:   return sphere_hit(&o->g.s, r, I, out);
  case ORC_GEOM_PLANE:    return 0; /* core/plane.cuh:25-29: stub, never hits */
  case ORC_GEOM_TRIANGLE: return triangle_hit(&o->g.t, r, I, out);
  default: return 0;
  }
}

/* ------------------------------------------------------------------ random numbers */

/*
 * The reference draws from cuRAND's default generator (curandState = XORWOW), one state per pixel set up by
 * curand_init(1984, pixel_index, 0, &state) (cuda_utils/curand_state.cuh:15-23).  cuRAND is a closed library
 * that is absent here; this restates its PUBLISHED algorithm (curand_kernel.h, CUDA toolkit):
 *   - Marsaglia's xorwow: five 32-bit xorshift words v[0..4] and a Weyl counter d stepping by 362437,
 *     output v[4] + d;
 *   - seeding: s0 = lo32(seed) ^ 0xaad26b49, s1 = hi32(seed) ^ 0xf7dcefdd, t0 = 1099087573 s0,
 *     t1 = 2591861531 s1; d = 6615241 + t1 + t0; v = {123456789 + t0, 362436069 ^ t0, 521288629 + t1,
 *     88675123 ^ t1, 5783321 + t0};
 *   - subsequence n starts n * 2^67 steps into the stream.  The xorshift part is linear over GF(2), so the
 *     jump is the 160x160 bit matrix A^(n 2^67) applied to v; d moves by 362437 n 2^67 = 0 (mod 2^32);
 *   - curand_uniform: x * 2^-32 + 2^-33 in fp32, x converted to float first: (0, 1].  (2^-32 is a power of
 *     two, so the product is exact and a fused multiply-add gives the same float.)
 * No cuRAND output vector exists in this image: parity with cuRAND's streams is UNPINNED.  What the tests
 * pin: A^(2^67) computed here by squaring equals rocRAND's precomputed table (rocrand_xorwow_precomputed.h,
 * a second implementation of the same recurrence and jump), small jumps equal stepping, and the GPU path
 * equals this restatement bit for bit.
 */
typedef struct { uint32_t v[5]; uint32_t d; } xorwow_t;

static inline uint32_t xorwow_next(xorwow_t *g)
{
  uint32_t t = g->v[0] ^ (g->v[0] >> 2);
  g->v[0] = g->v[1]; g->v[1] = g->v[2]; g->v[2] = g->v[3]; g->v[3] = g->v[4];
  g->v[4] = (g->v[4] ^ (g->v[4] << 4)) ^ (t ^ (t << 1));
  g->d += 362437u;
  return g->v[4] + g->d;
}

static inline float xorwow_uniform(xorwow_t *g)
{
  return (float)xorwow_next(g) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

/* bit matrices: row (w*32 + j) = image of the state with only bit j of word w set */
typedef struct { uint32_t row[160][5]; } bitmat_t;

static void bitmat_apply(const bitmat_t *m, const uint32_t in[5], uint32_t out[5])
{
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int w = 0; w < 5; ++w)
    for (int j = 0; j < 32; ++j)
      if ((in[w] >> j) & 1u)
        for (int k = 0; k < 5; ++k) r[k] ^= m->row[w * 32 + j][k];
  memcpy(out, r, sizeof r);
}

static void bitmat_step(bitmat_t *m) /* A: one xorwow step */
{
  for (int w = 0; w < 5; ++w)
    for (int j = 0; j < 32; ++j) {
      xorwow_t g;
      memset(&g, 0, sizeof g);
      g.v[w] = 1u << j;
      (void)xorwow_next(&g);
      memcpy(m->row[w * 32 + j], g.v, sizeof g.v);
    }
}

static void bitmat_square(bitmat_t *m)
{
  bitmat_t *sq = (bitmat_t *)malloc(sizeof *sq);
  for (int r = 0; r < 160; ++r) bitmat_apply(m, m->row[r], sq->row[r]);
  *m = *sq;
  free(sq);
}

/* A^(2^log2) */
static void bitmat_pow2(int log2, bitmat_t *m)
{
  bitmat_step(m);
  for (int i = 0; i < log2; ++i) bitmat_square(m);
}

static bitmat_t *g_seq_jump;     /* [64]: A^(2^(67+b)) */
static pthread_once_t g_seq_once = PTHREAD_ONCE_INIT;
static void seq_jump_build(void)
{
  g_seq_jump = (bitmat_t *)malloc(64 * sizeof(bitmat_t));
  bitmat_pow2(67, &g_seq_jump[0]);
  for (int b = 1; b < 64; ++b) { g_seq_jump[b] = g_seq_jump[b - 1]; bitmat_square(&g_seq_jump[b]); }
}

/* curand_init(seed, subsequence, 0, state) */
static void xorwow_init(uint64_t seed, uint64_t subsequence, xorwow_t *g)
{
  uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
  g->d = 6615241u + t1 + t0;
  g->v[0] = 123456789u + t0;
  g->v[1] = 362436069u ^ t0;
  g->v[2] = 521288629u + t1;
  g->v[3] = 88675123u ^ t1;
  g->v[4] = 5783321u + t0;
  pthread_once(&g_seq_once, seq_jump_build);
  for (int b = 0; b < 64; ++b)
    if ((subsequence >> b) & 1u) bitmat_apply(&g_seq_jump[b], g->v, g->v);
}

void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint32_t state6[6])
{
  xorwow_t g;
  xorwow_init(seed, subsequence, &g);
  memcpy(state6, g.v, sizeof g.v);
  state6[5] = g.d;
}

void orc_xorwow_draw(uint32_t state6[6], int n, uint32_t *raw, float *uniform)
{
  xorwow_t g;
  memcpy(g.v, state6, sizeof g.v);
  g.d = state6[5];
  for (int i = 0; i < n; ++i) {
    xorwow_t before = g;
    uint32_t x = xorwow_next(&g);
    if (raw) raw[i] = x;
    if (uniform) { g = before; uniform[i] = xorwow_uniform(&g); }
  }
  memcpy(state6, g.v, sizeof g.v);
  state6[5] = g.d;
}

void orc_xorwow_pow2_matrix(int log2, uint32_t out800[800])
{
  bitmat_t *m = (bitmat_t *)malloc(sizeof *m);
  bitmat_pow2(log2, m);
  memcpy(out800, m->row, sizeof m->row);
  free(m);
}

/* cuda_utils/curand_state.cuh:85-90: three draws (taken in x, y, z order -- C++ leaves the order of
 * constructor arguments to the compiler; unpinned like the streams themselves) */
static v3 random_vector(xorwow_t *g)
{
  float x = xorwow_uniform(g);
  float y = xorwow_uniform(g);
  float z = xorwow_uniform(g);
  return v3_make(x, y, z);
}

/* cuda_utils/curand_state.cuh:93-109.  The vector is tested against the unit ball around (0.5,0.5,0.5) but
 * returned UNcentred, in (0,1]^3; the test can never fail there (at most 0.75 <= 1), so a bounce costs
 * exactly three draws.  The loop is kept as the reference has it. */
static v3 random_in_unit_sphere(xorwow_t *g)
{
  const v3 centre = v3_make(0.5f, 0.5f, 0.5f);
  v3 v = random_vector(g);
  for (;;) {
    v3 vc = v3_sub(v, centre);
    if (v3_dot(vc, vc) <= 1.0f) break;
    v = random_vector(g);
  }
  return v;
}

/* ------------------------------------------------------------------ materials */

/* core/material.cuh:151-158 (direction only; origin is irrelevant to callers here) */
static inline v3 metallic_reflect(v3 incoming_dir, v3 normal)
{
  float s = 2.0f * v3_dot(incoming_dir, normal);
  return v3_sub(incoming_dir, v3_scale(s, normal));
}

/* core/scene_object.cuh:61-78 -> core/material.cuh:63-75 / :134-147 */
static v3 object_light_bounce(const object_t *o, const ray_t *incoming_light, v3 light_color, v3 normal,
                              const ray_t *viewing)
{
  if (o->mat_kind == ORC_MAT_DIFFUSE) {
    float scale = -v3_dot(incoming_light->d, normal);
    if (scale > 0.0f) return v3_scale(scale, light_color);
    return v3_make(0.0f, 0.0f, 0.0f);
  }
  v3 refl = metallic_reflect(incoming_light->d, normal);
  if (v3_dot(refl, viewing->d) < -0.99f) return light_color;
  return v3_make(0.001f, 0.001f, 0.001f);
}

/* core/material.cuh:17-24 */
static inline v3 material_color_fn(v3 mult, v3 c) { return v3_make(c.x * mult.x, c.y * mult.y, c.z * mult.z); }

/* ------------------------------------------------------------------ lights */

typedef struct { ray_t ray; hit_t hit; v3 color; } plr_t; /* core/light.cuh:23-34 */

/* core/light.cuh:73-81 (directional) and :47-53 (point) */
static plr_t light_ray_from(const light_t *l, v3 p)
{
  plr_t r;
  if (l->kind == ORC_LIGHT_DIRECTIONAL) {
    r.ray.o = p;
    r.ray.d = v3_neg(l->v);
    r.hit.t = INFINITY;
    r.hit.point = v3_make(INFINITY, INFINITY, INFINITY);
    r.hit.normal = l->v;
  } else {
    r.ray.o = p;
    r.ray.d = v3_sub(l->v, p);
    r.hit.t = 1.0f;
    r.hit.point = l->v;
    r.hit.normal = v3_normalized(v3_sub(p, l->v));
  }
  r.color = l->color;
  return r;
}

/* core/light.cuh:135-178.  Returns 1 when at least one light contributed. */
static int light_color(const orc_scene *s, const hit_t *hit, int hit_index, const ray_t *viewing, v3 *result,
                       orc_stats *st)
{
  int have = 0;
  for (int li = 0; li < s->nlight; ++li) {
    plr_t plr = light_ray_from(&s->light[li], hit->point);
    if (v3_dot(plr.ray.d, hit->normal) < 0) continue;
    st->shadow_rays++;

    int obstructed = 0;
    interval_t I = {1e-3f, plr.hit.t, 0, 0}; /* core/light.cuh:30-31 */
    for (int oi = 0; oi < s->nobj; ++oi) {
      hit_t other;
      if (object_hit(&s->obj[oi], &plr.ray, I, &other)) { obstructed = 1; break; }
    }
    if (obstructed) { st->occluded++; continue; }
    st->lit++;

    /* the reference passes the LIGHT's hit point/normal here (core/light.cuh:164-168) */
    v3 color = object_light_bounce(&s->obj[hit_index], &plr.ray, plr.color, plr.hit.normal, viewing);
    if (have) *result = v3_add(*result, color);
    else { *result = color; have = 1; }
  }
  return have;
}

/* ------------------------------------------------------------------ color_ray */

/* src/main.cu:44-105.  max_depth plays config::MAX_RECURSION_DEPTH.
 * *id_out / *t_out get the depth-0 closest hit. */
static v3 color_ray(const orc_scene *s, const ray_t *r, int depth, int max_depth, int32_t *id_out, float *t_out,
                    orc_stats *st, xorwow_t *rng)
{
  if (depth > max_depth) return v3_make(0.01f, 0.01f, 0.01f);

  interval_t I = {0.01f, INFINITY, 0, 0};
  float lowest_t = INFINITY;
  int hit_index = 0, have_hit = 0;
  hit_t best;
  memset(&best, 0, sizeof best);

  for (int i = 0; i < s->nobj; ++i) {
    hit_t h;
    if (object_hit(&s->obj[i], r, I, &h)) {
      if (h.t < lowest_t) { lowest_t = h.t; best = h; hit_index = i; have_hit = 1; }
    }
  }
  if (depth == 0) {
    if (id_out) *id_out = have_hit ? hit_index : -1;
    if (t_out) *t_out = lowest_t;
  }

  if (have_hit) {
    if (depth == 0) st->hit_pixels++;
    const object_t *o = &s->obj[hit_index];
    /* src/main.cu:80-83: the bounce comes first and a diffuse one advances the pixel's random stream
     * (core/material.cuh:52-61), also at the last level, whose bounce ray is never followed.  With
     * max_depth == 0 no level is ever followed and no stream is set up: rng is NULL there. */
    ray_t bounce;
    bounce.o = best.point;
    if (o->mat_kind == ORC_MAT_DIFFUSE) {
      if (rng) bounce.d = v3_add(best.normal, random_in_unit_sphere(rng));
      else bounce.d = best.normal;
    } else {
      bounce.d = metallic_reflect(r->d, best.normal); /* core/material.cuh:126-132 */
    }
    v3 light = v3_make(0, 0, 0);
    int have_light = light_color(s, &best, hit_index, r, &light, st);

    /* beyond the limit the recursive call returns 0.01 at once, src/main.cu:51-53 */
    v3 incoming = color_ray(s, &bounce, depth + 1, max_depth, 0, 0, st, rng);
    if (have_light) return v3_add(material_color_fn(o->color, incoming), material_color_fn(o->color, light));
    return material_color_fn(o->color, incoming);
  }

  /* background, src/main.cu:101-104 */
  v3 dir = v3_normalized(r->d);
  float y = 0.5f * (dir.y + 1.0f);
  return v3_add(v3_scale(1.0f - y, v3_make(1.0f, 1.0f, 1.0f)), v3_scale(y, v3_make(0.5f, 0.7f, 1.0f)));
}

/* ------------------------------------------------------------------ camera + samples */

/* pipeline/viewport.cuh:9-28 */
void orc_viewport(const float look_from[3], const float look_at[3], const float up[3], float vfov_deg, float aspect,
                  float cam12[12])
{
  v3 from = v3_from(look_from), at = v3_from(look_at), upv = v3_from(up);
  float theta = (float)(vfov_deg * M_PI / 180.0f); /* float*double/float -> double, stored to float */
  float half_height = tanf(theta / 2);
  float half_width = aspect * half_height;
  v3 w = v3_normalized(v3_sub(from, at));
  v3 u = v3_normalized(v3_cross(upv, w));
  v3 v = v3_cross(w, u);
  v3 up_left = v3_sub(v3_sub(v3_sub(from, v3_scale(half_width, u)), v3_scale(half_height, v)), w);
  v3 horizontal = v3_scale(2.0f * half_width, u);
  v3 vertical = v3_scale(2.0f * half_height, v);
  v3_to(from, cam12 + 0);
  v3_to(horizontal, cam12 + 3);
  v3_to(vertical, cam12 + 6);
  v3_to(up_left, cam12 + 9);
}

/* pipeline/viewport.cuh:30-34 */
static inline ray_t compute_ray(const float cam[12], float h, float v)
{
  v3 o = v3_from(cam + 0), hor = v3_from(cam + 3), ver = v3_from(cam + 6), ul = v3_from(cam + 9);
  ray_t r;
  r.o = o;
  r.d = v3_sub(v3_add(v3_add(ul, v3_scale(h, hor)), v3_scale(v, ver)), o);
  return r;
}

#define MS_W 11
#define MS_H 11
/* pipeline/multisample_mesh.cuh:17-31 */
static void msmesh_offsets(float norm_x, float norm_y, float off[MS_W * MS_H][2])
{
  const float DIV_H = (float)(MS_H + 2), DIV_W = (float)(MS_W + 2);
  for (int r = 0; r < MS_H; ++r) {
    float y = (-1.0f + 2.0f * (float)(r + 1) / DIV_H) / norm_y;
    for (int c = 0; c < MS_W; ++c) {
      float x = (-1.0f + 2.0f * (float)(c + 1) / DIV_W) / norm_x;
      off[r * MS_W + c][0] = x;
      off[r * MS_W + c][1] = y;
    }
  }
}

/* ------------------------------------------------------------------ render */

typedef struct {
  const orc_scene *s; const float *cam; int W, H, row_begin, row_end, spp, max_depth, any_diffuse;
  float *rgb; int32_t *id; float *t; orc_stats st;
} job_t;

/* src/main.cu:132-156, one row range */
static void render_rows(job_t *j)
{
  float off[MS_W * MS_H][2];
  int nsamp = 1;
  if (j->spp == MS_W * MS_H) { msmesh_offsets((float)j->W, (float)j->H, off); nsamp = MS_W * MS_H; }
  memset(&j->st, 0, sizeof j->st);

  for (int row = j->row_begin; row < j->row_end; ++row) {
    for (int col = 0; col < j->W; ++col) {
      size_t pix = (size_t)row * j->W + col;
      float u = (float)col / (float)j->W;
      float v = (float)row / (float)j->H;
      v3 acc = v3_make(0.0f, 0.0f, 0.0f);
      int32_t id = -1; float t = INFINITY;
      /* src/main.cu:133-135: a private copy of the pixel's state, shared by all of its samples */
      xorwow_t rng_state, *rng = 0;
      if (j->max_depth > 0 && j->any_diffuse) { xorwow_init(j->s->rng_seed, (uint64_t)pix, &rng_state); rng = &rng_state; }
      for (int sidx = 0; sidx < nsamp; ++sidx) {
        float su = u, sv = v;
        if (nsamp > 1) { su = u + off[sidx][0]; sv = v + off[sidx][1]; }
        ray_t ray = compute_ray(j->cam, su, sv);
        j->st.primary_rays++;
        acc = v3_add(acc, color_ray(j->s, &ray, 0, j->max_depth, &id, &t, &j->st, rng));
      }
      acc = v3_div(acc, (float)nsamp);
      if (j->rgb) v3_to(acc, j->rgb + 3 * pix);
      if (j->id) j->id[pix] = id;
      if (j->t) j->t[pix] = t;
    }
  }
}

static void *render_thread(void *p) { render_rows((job_t *)p); return 0; }

int orc_render(const orc_scene *s, const float cam12[12], int W, int H, int row_begin, int row_end, int spp,
               int max_depth, int nthreads, float *rgb, int32_t *id, float *t, orc_stats *stats)
{
  if (!s || !cam12 || W <= 0 || H <= 0 || row_begin < 0 || row_end > H || row_begin > row_end) return -1;
  if (spp != 1 && spp != MS_W * MS_H) return -1;
  if (max_depth < 0) return -1;
  int any_diffuse = 0;
  for (int i = 0; i < s->nobj; ++i)
    if (s->obj[i].mat_kind == ORC_MAT_DIFFUSE) any_diffuse = 1;
  if (nthreads < 1) nthreads = 1;
  int rows = row_end - row_begin;
  if (nthreads > rows) nthreads = rows > 0 ? rows : 1;

  job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
  pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
  if (!jobs || !th) { free(jobs); free(th); return -3; }
  /* rows are dealt in contiguous blocks of 8 round-robin so sky and geometry rows mix */
  /* (simple static split: contiguous ranges; balance is not the point of a baseline) */
  for (int k = 0; k < nthreads; ++k) {
    jobs[k].s = s; jobs[k].cam = cam12; jobs[k].W = W; jobs[k].H = H;
    jobs[k].row_begin = row_begin + (int)((long long)rows * k / nthreads);
    jobs[k].row_end = row_begin + (int)((long long)rows * (k + 1) / nthreads);
    jobs[k].spp = spp; jobs[k].max_depth = max_depth; jobs[k].any_diffuse = any_diffuse;
    jobs[k].rgb = rgb; jobs[k].id = id; jobs[k].t = t;
  }
  if (nthreads == 1) {
    render_rows(&jobs[0]);
  } else {
    for (int k = 0; k < nthreads; ++k) pthread_create(&th[k], 0, render_thread, &jobs[k]);
    for (int k = 0; k < nthreads; ++k) pthread_join(th[k], 0);
  }
  if (stats) {
    memset(stats, 0, sizeof *stats);
    for (int k = 0; k < nthreads; ++k) {
      stats->primary_rays += jobs[k].st.primary_rays;
      stats->shadow_rays += jobs[k].st.shadow_rays;
      stats->hit_pixels += jobs[k].st.hit_pixels;
      stats->occluded += jobs[k].st.occluded;
      stats->lit += jobs[k].st.lit;
    }
  }
  free(jobs); free(th);
  return 0;
}

/* ------------------------------------------------------------------ scene building */

orc_scene *orc_scene_create(void)
{
  orc_scene *s = (orc_scene *)calloc(1, sizeof(orc_scene));
  if (s) s->rng_seed = 1984; /* cuda_utils/curand_state.cuh:22 */
  return s;
}

void orc_scene_set_rng_seed(orc_scene *s, uint64_t seed) { s->rng_seed = seed; }

void orc_scene_destroy(orc_scene *s)
{
  if (!s) return;
  free(s->obj); free(s->light); free(s);
}

int orc_scene_num_objects(const orc_scene *s) { return s->nobj; }
int orc_scene_num_lights(const orc_scene *s) { return s->nlight; }

static object_t *push_object(orc_scene *s)
{
  if (s->nobj == s->cap_obj) {
    int nc = s->cap_obj ? s->cap_obj * 2 : 16;
    object_t *p = (object_t *)realloc(s->obj, (size_t)nc * sizeof(object_t));
    if (!p) return 0;
    s->obj = p; s->cap_obj = nc;
  }
  object_t *o = &s->obj[s->nobj++];
  memset(o, 0, sizeof *o);
  return o;
}

static light_t *push_light(orc_scene *s)
{
  if (s->nlight == s->cap_light) {
    int nc = s->cap_light ? s->cap_light * 2 : 8;
    light_t *p = (light_t *)realloc(s->light, (size_t)nc * sizeof(light_t));
    if (!p) return 0;
    s->light = p; s->cap_light = nc;
  }
  return &s->light[s->nlight++];
}

int orc_add_sphere(orc_scene *s, const float c[3], float r, int mat_kind, const float rgb[3])
{
  object_t *o = push_object(s);
  if (!o) return -1;
  o->geom_kind = ORC_GEOM_SPHERE; o->g.s.c = v3_from(c); o->g.s.r = r;
  o->mat_kind = mat_kind; o->color = v3_from(rgb);
  return s->nobj - 1;
}

int orc_add_triangle(orc_scene *s, const float a[3], const float b[3], const float c[3], int mat_kind,
                     const float rgb[3])
{
  object_t *o = push_object(s);
  if (!o) return -1;
  o->geom_kind = ORC_GEOM_TRIANGLE; o->g.t = triangle_make(v3_from(a), v3_from(b), v3_from(c));
  o->mat_kind = mat_kind; o->color = v3_from(rgb);
  return s->nobj - 1;
}

int orc_add_plane(orc_scene *s, float intercept, const float n[3], int mat_kind, const float rgb[3])
{
  object_t *o = push_object(s);
  if (!o) return -1;
  o->geom_kind = ORC_GEOM_PLANE; o->g.p.k = intercept; o->g.p.n = v3_from(n);
  o->mat_kind = mat_kind; o->color = v3_from(rgb);
  return s->nobj - 1;
}

int orc_add_objects(orc_scene *s, int n, const uint8_t *geom_kind, const float *geom9, const uint8_t *mat_kind,
                    const float *mat_rgb)
{
  for (int i = 0; i < n; ++i) {
    const float *g = geom9 + 9 * (size_t)i;
    int rc;
    if (geom_kind[i] == ORC_GEOM_SPHERE) rc = orc_add_sphere(s, g, g[3], mat_kind[i], mat_rgb + 3 * (size_t)i);
    else if (geom_kind[i] == ORC_GEOM_TRIANGLE) rc = orc_add_triangle(s, g, g + 3, g + 6, mat_kind[i], mat_rgb + 3 * (size_t)i);
    else rc = orc_add_plane(s, g[3], g, mat_kind[i], mat_rgb + 3 * (size_t)i);
    if (rc < 0) return rc;
  }
  return 0;
}

/* core/light.cuh:64-67: the constructor normalises the direction */
int orc_add_directional_light(orc_scene *s, const float dir[3], const float rgb[3])
{
  light_t *l = push_light(s);
  if (!l) return -1;
  l->kind = ORC_LIGHT_DIRECTIONAL; l->v = v3_normalized(v3_from(dir)); l->color = v3_from(rgb);
  return s->nlight - 1;
}

/* core/light.cuh:38-41 */
int orc_add_point_light(orc_scene *s, const float pos[3], const float rgb[3])
{
  light_t *l = push_light(s);
  if (!l) return -1;
  l->kind = ORC_LIGHT_POINT; l->v = v3_from(pos); l->color = v3_from(rgb);
  return s->nlight - 1;
}

static void add_sphere_v(orc_scene *s, v3 c, float r, int mk, v3 col)
{
  float cc[3], rgb[3]; v3_to(c, cc); v3_to(col, rgb);
  orc_add_sphere(s, cc, r, mk, rgb);
}
static void add_tri_v(orc_scene *s, v3 a, v3 b, v3 c, int mk, v3 col)
{
  float aa[3], bb[3], cc[3], rgb[3]; v3_to(a, aa); v3_to(b, bb); v3_to(c, cc); v3_to(col, rgb);
  orc_add_triangle(s, aa, bb, cc, mk, rgb);
}

/* src/main.cu:183-237; colour factories core/material.cuh:81-114, :163-196 */
void orc_build_default_scene(orc_scene *s)
{
  const v3 diffuse_red = v3_make(1.0f / 16.0f, 0.0f, 0.0f);
  const v3 diffuse_green = v3_make(0.0f, 1.0f / 16.0f, 0.0f);
  const v3 diffuse_blue = v3_make(0.0f, 0.0f, 1.0f / 16.0f);
  const v3 diffuse_yellow = v3_make(1.0f / 16.0f, 1.0f / 16.0f, 0.0f);
  const v3 diffuse_black = v3_make(0.001f, 0.001f, 0.001f);
  const v3 metallic_white = v3_make(1.0f, 1.0f, 1.0f);

  v3 line_start = v3_make(-2.25f, -1.0f, -10.f);
  v3 line_dir = v3_make(2.25f, 1.0f, -3.0f);
  v3 line_up = v3_make(0.0f, 1.0f, 0.0f);
  v3 line_dir_normed = v3_normalized(line_dir);
  line_up = v3_normalized(v3_sub(line_up, v3_scale(v3_dot(line_up, line_dir_normed), line_dir_normed)));
  v3 line_horizontal = v3_cross(line_up, line_dir_normed);

  v3 t1_center = v3_add(line_start, v3_scale(0.6f, line_dir)); /* 0.6 (double literal) acts as 0.6f */
  v3 t1_a = v3_add(t1_center, v3_scale(0.5f, line_up));
  v3 t1_b = v3_sub(v3_add(t1_center, v3_scale(0.5f, line_horizontal)), v3_scale(0.5f / 1.5f, line_up));
  v3 t1_c = v3_sub(v3_sub(t1_center, v3_scale(0.5f, line_horizontal)), v3_scale(0.5f / 1.5f, line_up));

  v3 t2_center = v3_add(line_start, v3_scale(1.1f, line_dir));
  v3 t2_a = v3_add(t2_center, v3_scale(0.75f, line_up));
  v3 t2_b = v3_sub(v3_add(t2_center, v3_scale(0.75f, line_horizontal)), v3_scale(0.75f / 1.5f, line_up));
  v3 t2_c = v3_sub(v3_sub(t2_center, v3_scale(0.75f, line_horizontal)), v3_scale(0.75f / 1.5f, line_up));

  v3 mirror_ul = v3_make(-4.5f, 2.0f, -21.0f), mirror_dl = v3_make(-4.5f, -2.0f, -21.0f);
  v3 mirror_ur = v3_make(4.5f, 2.0f, -21.0f), mirror_dr = v3_make(4.5f, -2.0f, -21.0f);
  v3 frame_ul = v3_make(-4.6f, 2.1f, -21.1f), frame_dl = v3_make(-4.6f, -2.1f, -21.1f);
  v3 frame_ur = v3_make(4.6f, 2.1f, -21.1f), frame_dr = v3_make(4.6f, -2.1f, -21.1f);

  add_sphere_v(s, v3_add(line_start, v3_scale(0.0f, line_dir)), 0.25f, ORC_MAT_DIFFUSE, diffuse_red);
  add_tri_v(s, t1_a, t1_b, t1_c, ORC_MAT_DIFFUSE, diffuse_yellow);
  add_sphere_v(s, v3_add(line_start, v3_scale(0.8f, line_dir)), 0.5f, ORC_MAT_DIFFUSE, diffuse_green);
  add_tri_v(s, t2_a, t2_b, t2_c, ORC_MAT_DIFFUSE, diffuse_yellow);
  add_sphere_v(s, v3_add(line_start, v3_scale(2.0f, line_dir)), 1.0f, ORC_MAT_DIFFUSE, diffuse_blue);
  add_tri_v(s, mirror_ul, mirror_ur, mirror_dl, ORC_MAT_METALLIC, metallic_white);
  add_tri_v(s, mirror_ur, mirror_dr, mirror_dl, ORC_MAT_METALLIC, metallic_white);
  add_tri_v(s, frame_ul, frame_ur, frame_dl, ORC_MAT_DIFFUSE, diffuse_black);
  add_tri_v(s, frame_ur, frame_dr, frame_dl, ORC_MAT_DIFFUSE, diffuse_black);
  add_sphere_v(s, v3_make(-4.5f, 0.0f, -13.0f), 1.0f, ORC_MAT_METALLIC, metallic_white);

  float ld[3], lc[3] = {16.0f, 16.0f, 16.0f};
  v3_to(line_dir, ld);
  orc_add_directional_light(s, ld, lc);
}

/* ------------------------------------------------------------------ single-primitive entry points */

void orc_triangle_precompute(const float a[3], const float b[3], const float c[3], float out16[16])
{
  triangle_t t = triangle_make(v3_from(a), v3_from(b), v3_from(c));
  v3_to(t.pn, out16 + 0);   out16[3] = t.kp;
  v3_to(t.ab_c.n, out16 + 4);  out16[7] = t.ab_c.k;
  v3_to(t.bc_a.n, out16 + 8);  out16[11] = t.bc_a.k;
  v3_to(t.ca_b.n, out16 + 12); out16[15] = t.ca_b.k;
}

static int emit_hit(int ok, const hit_t *h, float *t, float point[3], float normal[3])
{
  if (ok) {
    if (t) *t = h->t;
    if (point) v3_to(h->point, point);
    if (normal) v3_to(h->normal, normal);
  }
  return ok;
}

int orc_sphere_hit(const float c[3], float r, const float ro[3], const float rd[3], float tmin, float tmax, float *t,
                   float point[3], float normal[3])
{
  sphere_t s = {v3_from(c), r};
  ray_t ray = {v3_from(ro), v3_from(rd)};
  interval_t I = {tmin, tmax, 0, 0};
  hit_t h;
  return emit_hit(sphere_hit(&s, &ray, I, &h), &h, t, point, normal);
}

int orc_triangle_hit(const float a[3], const float b[3], const float c[3], const float ro[3], const float rd[3],
                     float tmin, float tmax, float *t, float point[3], float normal[3])
{
  triangle_t tr = triangle_make(v3_from(a), v3_from(b), v3_from(c));
  ray_t ray = {v3_from(ro), v3_from(rd)};
  interval_t I = {tmin, tmax, 0, 0};
  hit_t h;
  return emit_hit(triangle_hit(&tr, &ray, I, &h), &h, t, point, normal);
}

/* the same with any of the four interval modes (core/interval.cuh:14-23): prim = centre xyz + radius (sphere) or the 16
 * precomputed floats (triangle) */
int orc_prim_hit_interval(int geom_kind, const float *prim, const float ro[3], const float rd[3], float tmin, float tmax,
                          int lower_closed, int upper_closed, float *t, float point[3], float normal[3])
{
  ray_t ray = {v3_from(ro), v3_from(rd)};
  interval_t I = {tmin, tmax, !lower_closed, !upper_closed};
  hit_t h;
  if (geom_kind == ORC_GEOM_SPHERE) {
    sphere_t s = {v3_from(prim), prim[3]};
    return emit_hit(sphere_hit(&s, &ray, I, &h), &h, t, point, normal);
  }
  if (geom_kind == ORC_GEOM_TRIANGLE) {
    triangle_t tr;
    memset(&tr, 0, sizeof tr);
    tr.pn = v3_from(prim + 0); tr.kp = prim[3];
    tr.ab_c.n = v3_from(prim + 4); tr.ab_c.k = prim[7];
    tr.bc_a.n = v3_from(prim + 8); tr.bc_a.k = prim[11];
    tr.ca_b.n = v3_from(prim + 12); tr.ca_b.k = prim[15];
    return emit_hit(triangle_hit(&tr, &ray, I, &h), &h, t, point, normal);
  }
  return emit_hit(0, &h, t, point, normal); /* core/plane.cuh:25-29 */
}

/* object i as it was added: kind, geometry (sphere: centre xyz, radius; triangle: a, b, c; plane: k, normal), material */
int orc_scene_get_object(const orc_scene *s, int i, int *geom_kind, float geom[9], int *mat_kind, float rgb[3])
{
  if (!s || i < 0 || i >= s->nobj) return -1;
  const object_t *o = &s->obj[i];
  memset(geom, 0, 9 * sizeof(float));
  *geom_kind = o->geom_kind;
  switch (o->geom_kind) {
  case ORC_GEOM_SPHERE: v3_to(o->g.s.c, geom); geom[3] = o->g.s.r; break;
  case ORC_GEOM_TRIANGLE: v3_to(o->g.t.a, geom); v3_to(o->g.t.b, geom + 3); v3_to(o->g.t.c, geom + 6); break;
  default: geom[0] = o->g.p.k; v3_to(o->g.p.n, geom + 1); break;
  }
  *mat_kind = o->mat_kind;
  v3_to(o->color, rgb);
  return 0;
}

/* core/sphere.cuh:26-30 (compute_quadratic_params :89-114): strictly positive discriminant */
int orc_sphere_can_hit(const float c[3], float r, const float ro[3], const float rd[3])
{
  v3 o = v3_from(ro), d = v3_from(rd);
  v3 oc = v3_sub(o, v3_from(c));
  float a = v3_dot(d, d);
  float b = v3_dot(oc, d);
  float cc = v3_dot(oc, oc) - r * r;
  float disc = b * b - a * cc;
  return disc > 0.0f;
}

/* core/triangle.cuh:68-75: the ray is not parallel to the triangle's plane */
int orc_triangle_can_hit(const float a[3], const float b[3], const float c[3], const float rd[3])
{
  triangle_t tr = triangle_make(v3_from(a), v3_from(b), v3_from(c));
  return v3_dot(v3_from(rd), tr.pn) != 0.0f;
}

int orc_triangle16_hit(const float q[16], const float ro[3], const float rd[3], float tmin, float tmax, float *t,
                       float point[3], float normal[3])
{
  triangle_t tr;
  memset(&tr, 0, sizeof tr);
  tr.pn = v3_from(q + 0); tr.kp = q[3];
  tr.ab_c.n = v3_from(q + 4); tr.ab_c.k = q[7];
  tr.bc_a.n = v3_from(q + 8); tr.bc_a.k = q[11];
  tr.ca_b.n = v3_from(q + 12); tr.ca_b.k = q[15];
  ray_t ray = {v3_from(ro), v3_from(rd)};
  interval_t I = {tmin, tmax, 0, 0};
  hit_t h;
  return emit_hit(triangle_hit(&tr, &ray, I, &h), &h, t, point, normal);
}
