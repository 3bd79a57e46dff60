/*
 * rt_oracle.h -- CPU oracle for the render hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (libmrt.so and the C++ host layer) never
 * links, loads or calls anything declared here.
 *
 * What it is: a plain-C restatement of the reference's per-pixel algorithm
 * (AoS scene objects, a per-object type switch, brute-force loops), with one
 * fixed fp32 operation order and FMA contraction disabled at build time
 * (-ffp-contract=off).  Every function cites the reference file:line it follows;
 * paths are relative to the reference checkout.
 *
 * Pinning status: the reference cannot be built in this image (it needs CUDA,
 * Eigen and stb, none of which exist here, and stand-in headers are not an
 * accepted way to build it).  The oracle is pinned by
 *   (1) the 11 known-answer cases of the reference's own tests
 *       (test/spheretest.cu, test/triangletest.cu; tests/golden/reference_known_answers.json),
 *   (2) the full-frame anchors recorded by the survey session in BASELINE.md section 2
 *       (object-ID histograms at 640x480 and 1920x1080, ray counts, mean RGB,
 *       corner pixel; tests/golden/survey_anchors.json).
 * Anything neither covers (point lights, synthetic scenes, 121-spp mode) is
 * "parity unpinned" against the real reference and pinned only oracle-vs-GPU.
 *
 * Third-party arithmetic: the reference does its vector math with Eigen
 * (deps/eigen, un-vendored submodule, no pinned version).  The order restated
 * here is Eigen's non-vectorised fixed-size path:
 *   dot(a,b)      = a0*b0 + (a1*b1 + a2*b2)
 *   cross(a,b)    = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
 *   normalized(v) = z = dot(v,v); z > 0 ? v / sqrtf(z) : v     (true division)
 *   expressions evaluate component-wise, left to right.
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_GEOM_SPHERE = 0, ORC_GEOM_PLANE = 1, ORC_GEOM_TRIANGLE = 2 }; /* core/scene_object.cuh:14 */
enum { ORC_MAT_DIFFUSE = 0, ORC_MAT_METALLIC = 1 };                      /* core/scene_object.cuh:16 */
enum { ORC_LIGHT_DIRECTIONAL = 0, ORC_LIGHT_POINT = 1 };                 /* core/light.cuh:36,60 */

typedef struct orc_scene orc_scene;

typedef struct orc_stats {
  uint64_t primary_rays;   /* W*H*samples */
  uint64_t shadow_rays;    /* lights passing the L.N >= 0 test (core/light.cuh:148) */
  uint64_t hit_pixels;     /* samples whose primary ray hit something */
  uint64_t occluded;       /* shadow rays that found an occluder */
  uint64_t lit;            /* shadow rays that reached the light */
} orc_stats;

orc_scene *orc_scene_create(void);
void orc_scene_destroy(orc_scene *s);
int orc_scene_num_objects(const orc_scene *s);
int orc_scene_num_lights(const orc_scene *s);

/* geometry + material in one call, like SceneObject(geometry, material) */
int orc_add_sphere(orc_scene *s, const float c[3], float r, int mat_kind, const float rgb[3]);
int orc_add_triangle(orc_scene *s, const float a[3], const float b[3], const float c[3], int mat_kind,
                     const float rgb[3]);
int orc_add_plane(orc_scene *s, float intercept, const float n[3], int mat_kind, const float rgb[3]);
int orc_add_directional_light(orc_scene *s, const float dir[3], const float rgb[3]);
int orc_add_point_light(orc_scene *s, const float pos[3], const float rgb[3]);

/* bulk insert of n objects in scene order: geom_kind[i] is ORC_GEOM_SPHERE (geom9 = cx,cy,cz,r,...)
 * or ORC_GEOM_TRIANGLE (geom9 = a.xyz,b.xyz,c.xyz); 9 floats per object either way */
int orc_add_objects(orc_scene *s, int n, const uint8_t *geom_kind, const float *geom9, const uint8_t *mat_kind,
                    const float *mat_rgb);

/* the scene and light hard-coded in src/main.cu:183-237 */
void orc_build_default_scene(orc_scene *s);

/* rt::Viewport constructor (pipeline/viewport.cuh:9-28): 12 floats
 * origin(3), horizontal(3), vertical(3), up_left(3) */
void orc_viewport(const float look_from[3], const float look_at[3], const float up[3], float vfov_deg,
                  float aspect, float cam12[12]);

/* precomputed triangle as the reference's Triangle ctor leaves it
 * (core/triangle.cuh:58-66): pn.xyz,kp, n0.xyz,k0, n1.xyz,k1, n2.xyz,k2 */
void orc_triangle_precompute(const float a[3], const float b[3], const float c[3], float out16[16]);

/* single-primitive hit tests, the body of the reference's hit_sphere / hit_triangle
 * test kernels (test/spheretest.cu:6-12, test/triangletest.cu:6-12) with a caller
 * supplied closed interval.  Return 1 on hit. */
int orc_sphere_hit(const float c[3], float r, const float ro[3], const float rd[3], float tmin, float tmax,
                   float *t, float point[3], float normal[3]);
int orc_triangle_hit(const float a[3], const float b[3], const float c[3], const float ro[3],
                     const float rd[3], float tmin, float tmax, float *t, float point[3], float normal[3]);
/* ... with any of the four modes of rt::Interval (core/interval.cuh:14-42); prim: 4 floats (sphere) / 16 (triangle) */
int orc_prim_hit_interval(int geom_kind, const float *prim, const float ro[3], const float rd[3], float tmin, float tmax,
                          int lower_closed, int upper_closed, float *t, float point[3], float normal[3]);
int orc_scene_get_object(const orc_scene *s, int i, int *geom_kind, float geom[9], int *mat_kind, float rgb[3]);
/* Sphere::can_hit (core/sphere.cuh:26-30) and Triangle::can_hit (core/triangle.cuh:68-75); unused by render */
int orc_sphere_can_hit(const float c[3], float r, const float ro[3], const float rd[3]);
int orc_triangle_can_hit(const float a[3], const float b[3], const float c[3], const float rd[3]);
int orc_triangle16_hit(const float tri16[16], const float ro[3], const float rd[3], float tmin, float tmax,
                       float *t, float point[3], float normal[3]);

/* Render rows [row_begin,row_end) of a W x H frame (src/main.cu:132-156 per pixel).
 * spp: 1 = single sample at (u,v), 121 = MultisampleMesh.
 * max_depth: value of config::MAX_RECURSION_DEPTH (src/main.cu:26), 0..8 here.  With max_depth > 0 a
 * diffuse hit bounces along normal + random vector drawn from the pixel's XORWOW stream
 * (seed orc_scene_set_rng_seed, default 1984; subsequence = row*W + col), a metallic one mirrors.
 * Outputs are full-frame arrays (W*H*3 / W*H / W*H), row 0 = v 0; any may be NULL.
 * id = object index of the closest hit of the (last) sample, -1 on miss; t = +inf on miss.
 * nthreads <= 1 runs on the calling thread. */
int orc_render(const orc_scene *s, const float cam12[12], int W, int H, int row_begin, int row_end, int spp,
               int max_depth, int nthreads, float *rgb, int32_t *id, float *t, orc_stats *stats);

/* --- the random streams of the diffuse bounce (cuda_utils/curand_state.cuh:15-23, :85-109): cuRAND's XORWOW
 * restated from its published algorithm; parity with cuRAND's own output is unpinned (see rt_oracle.c). */
void orc_scene_set_rng_seed(orc_scene *s, uint64_t seed);
/* curand_init(seed, subsequence, 0): state6 = v[0..4], d */
void orc_xorwow_init(uint64_t seed, uint64_t subsequence, uint32_t state6[6]);
/* n draws from state6 (advanced in place): raw 32-bit outputs and/or curand_uniform floats; either may be NULL */
void orc_xorwow_draw(uint32_t state6[6], int n, uint32_t *raw, float *uniform);
/* A^(2^log2) as 160 rows of 5 words, row (w*32 + j) = image of bit j of word w */
void orc_xorwow_pow2_matrix(int log2, uint32_t out800[800]);

#ifdef __cplusplus
}
#endif
#endif
