// mrt_kernels.hip -- hand-written gfx950 kernels of the render hot path.
//
// Build with -ffp-contract=off and WITHOUT fast-math: every expression below has the
// operation order of the CPU oracle (oracle/rt_oracle.c), which is the order of the
// reference's Eigen expressions.  IEEE '/' and sqrtf (hipcc's default correctly
// rounded expansions) are part of the contract.
//
// Mapping (DESIGN.md "Kernel"):
//   workgroup = 256 threads = 4 wave64;  a wave owns an 8-row x (8*P)-column pixel
//   tile: lane = ty*8 + tx holds the P adjacent pixels (tx*P .. tx*P+P-1, ty).  The
//   primitive under test is the SAME for all 64 lanes, so it is read from LDS as a
//   broadcast (one ds_read_b128 per 16 bytes of primitive) and reused for P rays.
//   Scene SoA streams in HBM are transposed into float4 records while they are
//   staged into LDS; ray-invariant terms of the primary rays (which share the
//   camera origin) are folded into the record during staging:
//     sphere   primary record: (o-c).xyz, (o-c).(o-c) - r*r        shadow record: c.xyz, r*r
//     triangle record: pn.xyz,kp | n0.xyz,k0 | n1.xyz,k1 | n2.xyz,k2   + primary num = kp - o.pn
//   These are the same fp32 operations the reference performs per test, done once.

#include "mrt_internal.h"
#include "mrt_device.h"

#include <cstdlib>

namespace mrt {

// ------------------------------------------------------------------------------------------
// LDS staging.  lds4 is the dynamic LDS viewed as float4.
// Layout:  [0, s_cap)            sphere primary records
//          [s_cap, 2 s_cap)      sphere shadow records
//          [2 s_cap, +4 t_cap)   triangle records (4 float4 each)
//          then t_cap floats     primary numerators kp - o.pn
// Chunks are padded to a multiple of the unroll factor with records that can never hit:
//   sphere primary (0,0,0,+inf): disc = b*b - a*inf is -inf or NaN, never >= 0
//   sphere shadow  (0,0,0,-inf): c = oc.oc + inf = +inf, same
//   triangle       pn = 0:       rdn == 0, rejected by the rdn != 0 test
// ------------------------------------------------------------------------------------------

constexpr int US = 8; // spheres per unrolled block (8 x 16 B prefetched per block)
constexpr int UT = 2; // triangles per unrolled block (2 x 68 B)

struct LdsPlan {
  float4 *sphP, *sphS, *tri;
  float *triNum;
};

__device__ __forceinline__ LdsPlan lds_plan(float4 *base, int s_cap, int t_cap)
{
  LdsPlan p;
  p.sphP = base;
  p.sphS = base + s_cap;
  p.tri = base + 2 * s_cap;
  p.triNum = reinterpret_cast<float *>(base + 2 * s_cap + 4 * t_cap);
  return p;
}

__device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// which: bit0 primary records, bit1 shadow records
__device__ __forceinline__ void stage_spheres(const DevScene &sc, const DevCamera &cam, const LdsPlan &L, int begin,
                                              int count, int which)
{
  const int padded = round_up(count, US);
  for (int i = threadIdx.x; i < padded; i += blockDim.x) {
    if (i < count) {
      int g = begin + i;
      float cx = sc.sph_cx[g], cy = sc.sph_cy[g], cz = sc.sph_cz[g], r = sc.sph_r[g];
      float rr = r * r;
      if (which & 1) {
        float ocx = cam.ox - cx, ocy = cam.oy - cy, ocz = cam.oz - cz;
        float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - rr;
        L.sphP[i] = make_float4(ocx, ocy, ocz, c);
      }
      if (which & 2) L.sphS[i] = make_float4(cx, cy, cz, rr);
    } else {
      if (which & 1) L.sphP[i] = make_float4(0.0f, 0.0f, 0.0f, MRT_INF);
      if (which & 2) L.sphS[i] = make_float4(0.0f, 0.0f, 0.0f, -MRT_INF);
    }
  }
}

__device__ __forceinline__ void stage_triangles(const DevScene &sc, const DevCamera &cam, const LdsPlan &L, int begin,
                                                int count)
{
  const int nt = sc.nt;
  const int padded = round_up(count, UT);
  for (int i = threadIdx.x; i < padded; i += blockDim.x) {
    float4 q0, q1, q2, q3;
    if (i < count) {
      const float *p = sc.tri + begin + i;
      q0 = make_float4(p[0], p[nt], p[2 * nt], p[3 * nt]);
      q1 = make_float4(p[4 * nt], p[5 * nt], p[6 * nt], p[7 * nt]);
      q2 = make_float4(p[8 * nt], p[9 * nt], p[10 * nt], p[11 * nt]);
      q3 = make_float4(p[12 * nt], p[13 * nt], p[14 * nt], p[15 * nt]);
    } else {
      q0 = q1 = q2 = q3 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    L.tri[4 * i + 0] = q0;
    L.tri[4 * i + 1] = q1;
    L.tri[4 * i + 2] = q2;
    L.tri[4 * i + 3] = q3;
    L.triNum[i] = q0.w - dot3(cam.ox, cam.oy, cam.oz, q0.x, q0.y, q0.z);
  }
}

// ------------------------------------------------------------------------------------------
// Inner loops.  Every loop walks one staged chunk in blocks of U primitives: the records of
// a block are read from LDS back to back (broadcast reads: all lanes use the same address), then the
// whole block is evaluated, so one LDS round trip is paid per U primitives and the other waves of the
// SIMD cover it.
// ------------------------------------------------------------------------------------------

struct Tri {
  float4 q0, q1, q2, q3;
  float num;
};

__device__ __forceinline__ Tri load_tri(const LdsPlan &L, int j)
{
  Tri t;
  t.q0 = L.tri[4 * j + 0]; t.q1 = L.tri[4 * j + 1]; t.q2 = L.tri[4 * j + 2]; t.q3 = L.tri[4 * j + 3];
  t.num = L.triNum[j];
  return t;
}

// closest hit over one chunk of sphere primary records (src/main.cu:60-77, core/sphere.cuh:32-73)
template <int P>
__device__ __forceinline__ void closest_spheres(const float4 *__restrict__ rec, int padded, int prim0,
                                                const int32_t *__restrict__ sph_obj, const float (&dx)[P],
                                                const float (&dy)[P], const float (&dz)[P], const float (&aa)[P],
                                                float (&best_t)[P], int (&best_obj)[P], int (&best_prim)[P])
{
  for (int j0 = 0; j0 < padded; j0 += US) {
    float4 cur[US];
#pragma unroll
    for (int k = 0; k < US; ++k) cur[k] = rec[j0 + k];
    float b[US][P], disc[US][P];
    float dmax = -MRT_INF; // max ignores NaN discriminants, which can never be hits
#pragma unroll
    for (int k = 0; k < US; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        b[k][p] = dot3(cur[k].x, cur[k].y, cur[k].z, dx[p], dy[p], dz[p]);
        disc[k][p] = b[k][p] * b[k][p] - aa[p] * cur[k].w;
        dmax = fmaxf(dmax, disc[k][p]);
      }
    }
    const bool cand = dmax >= 0.0f;
    if (cand) {
#pragma unroll
      for (int k = 0; k < US; ++k) {
        bool ck = false;
#pragma unroll
        for (int p = 0; p < P; ++p) ck |= (disc[k][p] >= 0.0f);
        if (ck) {
          const int obj = sph_obj[prim0 + j0 + k];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (disc[k][p] >= 0.0f) {
              float t;
              if (sphere_roots(aa[p], b[k][p], disc[k][p], 0.01f, MRT_INF, t)) {
                if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
                  best_t[p] = t; best_obj[p] = obj; best_prim[p] = prim0 + j0 + k;
                }
              }
            }
          }
        }
      }
    }
  }
}

// closest hit over one chunk of triangle records (core/triangle.cuh:77-103); the rays share the origin o
template <int P>
__device__ __forceinline__ void closest_triangles(const LdsPlan &L, int padded, int prim0,
                                                  const int32_t *__restrict__ tri_obj, float ox, float oy, float oz,
                                                  const float (&dx)[P], const float (&dy)[P], const float (&dz)[P],
                                                  float (&best_t)[P], int (&best_obj)[P], int (&best_prim)[P])
{
  for (int j0 = 0; j0 < padded; j0 += UT) {
    Tri cur[UT];
#pragma unroll
    for (int k = 0; k < UT; ++k) cur[k] = load_tri(L, j0 + k);
    float t[UT][P];
    bool hit[UT][P];
    bool any = false;
#pragma unroll
    for (int k = 0; k < UT; ++k) {
      const Tri &T = cur[k];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float rdn = dot3(dx[p], dy[p], dz[p], T.q0.x, T.q0.y, T.q0.z);
        t[k][p] = T.num / rdn;
        float px = ox + t[k][p] * dx[p], py = oy + t[k][p] * dy[p], pz = oz + t[k][p] * dz[p];
        float e0 = dot3(T.q1.x, T.q1.y, T.q1.z, px, py, pz) - T.q1.w;
        float e1 = dot3(T.q2.x, T.q2.y, T.q2.z, px, py, pz) - T.q2.w;
        float e2 = dot3(T.q3.x, T.q3.y, T.q3.z, px, py, pz) - T.q3.w;
        hit[k][p] = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t[k][p]) &
                    (t[k][p] <= MRT_INF);
        any |= hit[k][p];
      }
    }
    if (any) {
#pragma unroll
      for (int k = 0; k < UT; ++k) {
        bool hk = false;
#pragma unroll
        for (int p = 0; p < P; ++p) hk |= hit[k][p];
        if (hk) {
          const int obj = tri_obj[prim0 + j0 + k];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            if (hit[k][p] &&
                (t[k][p] < best_t[p] || (t[k][p] == best_t[p] && t[k][p] < MRT_INF && obj < best_obj[p]))) {
              best_t[p] = t[k][p]; best_obj[p] = obj; best_prim[p] = prim0 + j0 + k;
            }
          }
        }
      }
    }
  }
}

// Shadow-ray directions: one per wave for a directional light, one per lane for a point light.
template <int P> struct UniformDir {
  float x, y, z, a, tl;
  __device__ __forceinline__ float dx(int) const { return x; }
  __device__ __forceinline__ float dy(int) const { return y; }
  __device__ __forceinline__ float dz(int) const { return z; }
  __device__ __forceinline__ float da(int) const { return a; }
  __device__ __forceinline__ float tmax(int) const { return tl; }
};
template <int P> struct LaneDir {
  float x[P], y[P], z[P], a[P];
  __device__ __forceinline__ float dx(int p) const { return x[p]; }
  __device__ __forceinline__ float dy(int p) const { return y[p]; }
  __device__ __forceinline__ float dz(int p) const { return z[p]; }
  __device__ __forceinline__ float da(int p) const { return a[p]; }
  __device__ __forceinline__ float tmax(int) const { return 1.0f; } // PointLight: t_light = 1 (core/light.cuh:50)
};

// any-hit over one chunk of sphere shadow records, interval [1e-3, t_light] (core/light.cuh:28-33).
// Returns false once no lane of the wave has an undecided ray left (ballot early-out).
template <int P, class D>
__device__ __forceinline__ bool occlude_spheres(const float4 *__restrict__ rec, int padded, const D &d,
                                                const float (&hx)[P], const float (&hy)[P], const float (&hz)[P],
                                                bool (&open)[P])
{
  for (int j0 = 0; j0 < padded; j0 += US) {
    float4 cur[US];
#pragma unroll
    for (int k = 0; k < US; ++k) cur[k] = rec[j0 + k];
    float b[US][P], disc[US][P];
    float dmax[P];
#pragma unroll
    for (int p = 0; p < P; ++p) dmax[p] = -MRT_INF;
#pragma unroll
    for (int k = 0; k < US; ++k) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float ocx = hx[p] - cur[k].x, ocy = hy[p] - cur[k].y, ocz = hz[p] - cur[k].z;
        b[k][p] = dot3(ocx, ocy, ocz, d.dx(p), d.dy(p), d.dz(p));
        float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - cur[k].w;
        disc[k][p] = b[k][p] * b[k][p] - d.da(p) * c;
        dmax[p] = fmaxf(dmax[p], disc[k][p]);
      }
    }
    bool cand = false;
#pragma unroll
    for (int p = 0; p < P; ++p) cand |= open[p] && (dmax[p] >= 0.0f);
    if (cand) {
#pragma unroll
      for (int k = 0; k < US; ++k) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (open[p] && disc[k][p] >= 0.0f) {
            float t;
            if (sphere_roots(d.da(p), b[k][p], disc[k][p], 1e-3f, d.tmax(p), t)) open[p] = false;
          }
        }
      }
      bool still = false;
#pragma unroll
      for (int p = 0; p < P; ++p) still |= open[p];
      if (!__any(still)) return false;
    }
  }
  return true;
}

template <int P, class D>
__device__ __forceinline__ bool occlude_triangles(const LdsPlan &L, int padded, const D &d, const float (&hx)[P],
                                                  const float (&hy)[P], const float (&hz)[P], bool (&open)[P])
{
  for (int j0 = 0; j0 < padded; j0 += UT) {
    Tri cur[UT];
#pragma unroll
    for (int k = 0; k < UT; ++k) cur[k] = load_tri(L, j0 + k);
    bool blocked = false;
#pragma unroll
    for (int k = 0; k < UT; ++k) {
      const Tri &T = cur[k];
#pragma unroll
      for (int p = 0; p < P; ++p) {
        float rdn = dot3(d.dx(p), d.dy(p), d.dz(p), T.q0.x, T.q0.y, T.q0.z);
        float t = (T.q0.w - dot3(hx[p], hy[p], hz[p], T.q0.x, T.q0.y, T.q0.z)) / rdn;
        float px = hx[p] + t * d.dx(p), py = hy[p] + t * d.dy(p), pz = hz[p] + t * d.dz(p);
        float e0 = dot3(T.q1.x, T.q1.y, T.q1.z, px, py, pz) - T.q1.w;
        float e1 = dot3(T.q2.x, T.q2.y, T.q2.z, px, py, pz) - T.q2.w;
        float e2 = dot3(T.q3.x, T.q3.y, T.q3.z, px, py, pz) - T.q3.w;
        bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= d.tmax(p));
        if (open[p] && h) { open[p] = false; blocked = true; }
      }
    }
    if (__any(blocked)) {
      bool still = false;
#pragma unroll
      for (int p = 0; p < P; ++p) still |= open[p];
      if (!__any(still)) return false;
    }
  }
  return true;
}

// One light's occlusion pass over the whole scene.  Streaming scenes restage every chunk, so all
// waves of the workgroup walk the chunk loops together (barriers) even when they have nothing left.
template <int P, class D>
__device__ __forceinline__ void occlusion_pass(const RenderArgs &A, const LdsPlan &L, const D &d,
                                               const float (&hx)[P], const float (&hy)[P], const float (&hz)[P],
                                               bool (&open)[P])
{
  const DevScene &sc = A.sc;
  bool live = false;
#pragma unroll
  for (int p = 0; p < P; ++p) live |= open[p];
  bool wave_live = __any(live);
  for (int c0 = 0; c0 < sc.ns; c0 += A.s_cap) {
    const int cnt = min(A.s_cap, sc.ns - c0);
    if (!A.resident) {
      __syncthreads();
      stage_spheres(sc, A.cam, L, c0, cnt, 2);
      __syncthreads();
    }
    if (wave_live) wave_live = occlude_spheres<P>(L.sphS, round_up(cnt, US), d, hx, hy, hz, open);
  }
  for (int c0 = 0; c0 < sc.nt; c0 += A.t_cap) {
    const int cnt = min(A.t_cap, sc.nt - c0);
    if (!A.resident) {
      __syncthreads();
      stage_triangles(sc, A.cam, L, c0, cnt);
      __syncthreads();
    }
    if (wave_live) wave_live = occlude_triangles<P>(L, round_up(cnt, UT), d, hx, hy, hz, open);
  }
}

// ------------------------------------------------------------------------------------------
// render kernel
// ------------------------------------------------------------------------------------------

template <int P>
__global__ __launch_bounds__(256) void render_kernel(RenderArgs A)
{
  extern __shared__ float4 lds4[];
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const LdsPlan L = lds_plan(lds4, A.s_cap, A.t_cap);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int col0 = (blockIdx.x * 4 + wave) * (8 * P) + tx * P;
  const int lrow = blockIdx.y * 8 + ty;
  const int strip = lrow / A.strip_rows;
  const int grow = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow - strip * A.strip_rows);
  const bool row_ok = (lrow < A.local_rows) && (grow < A.H);

  const int ns = sc.ns, nt = sc.nt;

  if (A.resident) {
    stage_spheres(sc, cam, L, 0, ns, 3);
    stage_triangles(sc, cam, L, 0, nt);
    __syncthreads();
  }

  bool valid[P];
#pragma unroll
  for (int p = 0; p < P; ++p) valid[p] = row_ok && (col0 + p < A.W);
  const float v = float(grow) / float(A.H); // src/main.cu:137

  float accr[P], accg[P], accb[P];
  int out_id[P];
  float out_t[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { accr[p] = 0.0f; accg[p] = 0.0f; accb[p] = 0.0f; out_id[p] = -1; out_t[p] = MRT_INF; }

  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;

  const int nsamp = A.spp;
  const float DIVW = 13.0f, DIVH = 13.0f; // MultisampleMesh::WIDTH/HEIGHT + 2

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    // sample offsets: pipeline/multisample_mesh.cuh:23-30 (zero offset in 1-spp mode)
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) {
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }

    // ---- primary rays: pipeline/viewport.cuh:30-34 ----
    float dx[P], dy[P], dz[P], aa[P];
    float best_t[P];
    int best_obj[P], best_prim[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float su = float(col0 + p) / float(A.W), sv = v; // src/main.cu:136
      if (nsamp > 1) { su = su + offx; sv = v + offy; }
      dx[p] = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
      dy[p] = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      dz[p] = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
      aa[p] = dot3(dx[p], dy[p], dz[p], dx[p], dy[p], dz[p]);
      best_t[p] = MRT_INF;
      best_obj[p] = 0x7fffffff;
      best_prim[p] = -1;
    }

    // ---- closest hit, interval [0.01, inf] ----
    for (int c0 = 0; c0 < ns; c0 += A.s_cap) {
      const int cnt = min(A.s_cap, ns - c0);
      if (!A.resident) {
        __syncthreads();
        stage_spheres(sc, cam, L, c0, cnt, 1);
        __syncthreads();
      }
      closest_spheres<P>(L.sphP, round_up(cnt, US), c0, sc.sph_obj, dx, dy, dz, aa, best_t, best_obj, best_prim);
    }
    for (int c0 = 0; c0 < nt; c0 += A.t_cap) {
      const int cnt = min(A.t_cap, nt - c0);
      if (!A.resident) {
        __syncthreads();
        stage_triangles(sc, cam, L, c0, cnt);
        __syncthreads();
      }
      closest_triangles<P>(L, round_up(cnt, UT), ns + c0, sc.tri_obj - ns, cam.ox, cam.oy, cam.oz, dx, dy, dz, best_t,
                           best_obj, best_prim);
    }

    // ---- hit record of the winner: point, normal (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
    bool have_hit[P];
    float hpx[P], hpy[P], hpz[P], hnx[P], hny[P], hnz[P];
    bool any_hit = false;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      have_hit[p] = best_prim[p] >= 0;
      hpx[p] = hpy[p] = hpz[p] = 0.0f;
      hnx[p] = hny[p] = hnz[p] = 0.0f;
      if (have_hit[p]) {
        const float t = best_t[p];
        hpx[p] = cam.ox + t * dx[p]; hpy[p] = cam.oy + t * dy[p]; hpz[p] = cam.oz + t * dz[p];
        if (best_prim[p] < ns) {
          const int g = best_prim[p];
          float nx = hpx[p] - sc.sph_cx[g], ny = hpy[p] - sc.sph_cy[g], nz = hpz[p] - sc.sph_cz[g];
          normalize3(nx, ny, nz);
          hnx[p] = nx; hny[p] = ny; hnz[p] = nz;
        } else {
          const float *q = sc.tri + (best_prim[p] - ns);
          const float pnx = q[0], pny = q[nt], pnz = q[2 * nt];
          const float rdn = dot3(dx[p], dy[p], dz[p], pnx, pny, pnz);
          const bool front = rdn < 0;
          hnx[p] = front ? pnx : -pnx; hny[p] = front ? pny : -pny; hnz[p] = front ? pnz : -pnz;
        }
        if (valid[p]) n_hit++;
      }
      any_hit |= have_hit[p];
    }
    const bool wave_any_hit = __any(any_hit);

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr[P], lg[P], lb[P];
    bool have_light[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { lr[p] = lg[p] = lb[p] = 0.0f; have_light[p] = false; }

    // Resident scenes let a wave with no hit skip the light loop; streaming scenes cannot (barriers).
    if (!A.resident || wave_any_hit) {
      for (int li = 0; li < sc.nl; ++li) {
        const int lkind = sc.light_kind[li];
        const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
        const float lcr = sc.light_rgb[3 * li], lcg = sc.light_rgb[3 * li + 1], lcb = sc.light_rgb[3 * li + 2];
        bool need[P], open[P];
        float scale[P], refl_dot[P]; // inputs of light_bounce, see below

        if (lkind == 0) {
          // DirectionalLight::ray_from, core/light.cuh:73-81: ray {point, -dir}, t = inf, light normal = dir
          UniformDir<P> d;
          d.x = -lvx; d.y = -lvy; d.z = -lvz; d.tl = MRT_INF;
          d.a = dot3(d.x, d.y, d.z, d.x, d.y, d.z);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            // the light must not shine through the surface, core/light.cuh:148
            need[p] = have_hit[p] && !(dot3(d.x, d.y, d.z, hnx[p], hny[p], hnz[p]) < 0);
            open[p] = need[p];
          }
          occlusion_pass<P>(A, L, d, hpx, hpy, hpz, open);
          // light_bounce gets the LIGHT's normal (= dir), as the reference passes it (core/light.cuh:164-168)
          const float dn = dot3(d.x, d.y, d.z, lvx, lvy, lvz);
          const float s2 = 2.0f * dn;
          const float rx = d.x - s2 * lvx, ry = d.y - s2 * lvy, rz = d.z - s2 * lvz;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            scale[p] = -dn;
            refl_dot[p] = dot3(rx, ry, rz, dx[p], dy[p], dz[p]);
          }
        } else {
          // PointLight::ray_from, core/light.cuh:47-53: ray {point, pos - point}, t = 1,
          // light normal = (point - pos).normalized()
          LaneDir<P> d;
          float lnx[P], lny[P], lnz[P];
#pragma unroll
          for (int p = 0; p < P; ++p) {
            d.x[p] = lvx - hpx[p]; d.y[p] = lvy - hpy[p]; d.z[p] = lvz - hpz[p];
            d.a[p] = dot3(d.x[p], d.y[p], d.z[p], d.x[p], d.y[p], d.z[p]);
            float nx = hpx[p] - lvx, ny = hpy[p] - lvy, nz = hpz[p] - lvz;
            normalize3(nx, ny, nz);
            lnx[p] = nx; lny[p] = ny; lnz[p] = nz;
            need[p] = have_hit[p] && !(dot3(d.x[p], d.y[p], d.z[p], hnx[p], hny[p], hnz[p]) < 0);
            open[p] = need[p];
          }
          occlusion_pass<P>(A, L, d, hpx, hpy, hpz, open);
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const float dn = dot3(d.x[p], d.y[p], d.z[p], lnx[p], lny[p], lnz[p]);
            const float s2 = 2.0f * dn;
            const float rx = d.x[p] - s2 * lnx[p], ry = d.y[p] - s2 * lny[p], rz = d.z[p] - s2 * lnz[p];
            scale[p] = -dn;
            refl_dot[p] = dot3(rx, ry, rz, dx[p], dy[p], dz[p]);
          }
        }

#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (need[p] && valid[p]) { n_shadow++; if (!open[p]) n_occl++; }
          if (open[p]) {
            float cr, cg, cb;
            if (sc.mat_kind[best_obj[p]] == 0) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
              if (scale[p] > 0.0f) { cr = scale[p] * lcr; cg = scale[p] * lcg; cb = scale[p] * lcb; }
              else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
            } else {                             // MetallicMaterial::light_bounce, core/material.cuh:134-158
              if (refl_dot[p] < -0.99f) { cr = lcr; cg = lcg; cb = lcb; }
              else { cr = 0.001f; cg = 0.001f; cb = 0.001f; }
            }
            if (have_light[p]) { lr[p] += cr; lg[p] += cg; lb[p] += cb; }
            else { lr[p] = cr; lg[p] = cg; lb[p] = cb; have_light[p] = true; }
          }
        }
      }
    }

    // ---- combine: src/main.cu:91-104 with MAX_RECURSION_DEPTH = 0 ----
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float cr, cg, cb;
      if (have_hit[p]) {
        const float *m = sc.mat_rgb + 3 * best_obj[p];
        const float mr = m[0], mg = m[1], mb = m[2];
        const float inc = 0.01f; // what the depth-1 call returns, src/main.cu:51-53
        if (have_light[p]) {
          cr = inc * mr + lr[p] * mr; cg = inc * mg + lg[p] * mg; cb = inc * mb + lb[p] * mb;
        } else {
          cr = inc * mr; cg = inc * mg; cb = inc * mb;
        }
      } else {
        float nx = dx[p], ny = dy[p], nz = dz[p];
        normalize3(nx, ny, nz);
        float y = 0.5f * (ny + 1.0f);
        float w0 = 1.0f - y;
        cr = w0 * 1.0f + y * 0.5f; cg = w0 * 1.0f + y * 0.7f; cb = w0 * 1.0f + y * 1.0f;
      }
      accr[p] += cr; accg[p] += cg; accb[p] += cb; // src/main.cu:151
      out_id[p] = have_hit[p] ? best_obj[p] : -1;
      out_t[p] = best_t[p];
    }
  }

  // ---- write out: src/main.cu:155 ----
  const float fn = float(nsamp);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (valid[p]) {
      const size_t pix = size_t(A.frame_layout ? grow : lrow) * A.W + (col0 + p);
      A.rgb[3 * pix + 0] = accr[p] / fn;
      A.rgb[3 * pix + 1] = accg[p] / fn;
      A.rgb[3 * pix + 2] = accb[p] / fn;
      if (A.id) A.id[pix] = out_id[p];
      if (A.t) A.t[pix] = out_t[p];
    }
  }

  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * ((blockIdx.x + 7u * blockIdx.y) & (unsigned)(A.stats_slots - 1));
    // wave reduction, one atomic per wave (only on the stats path, never in timed loops)
    unsigned int npix = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) npix += valid[p] ? 1u : 0u;
    unsigned int v4[4] = {npix * (unsigned)nsamp, n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int x = v4[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[k], (unsigned long long)x);
    }
  }
}

static size_t lds_bytes(int s_cap, int t_cap) { return size_t(2 * s_cap + 4 * t_cap) * 16 + size_t(t_cap) * 4; }
static int round_up_h(int x, int m) { return (x + m - 1) / m * m; }

hipError_t launch_render(const RenderArgs &in, hipStream_t stream)
{
  RenderArgs A = in;
  const int ns = A.sc.ns, nt = A.sc.nt;
  // resident when the whole scene (both sphere records + triangle records) fits in 48 KiB
  const int ns_pad = round_up_h(ns > 0 ? ns : 1, US), nt_pad = round_up_h(nt > 0 ? nt : 1, UT);
  if (lds_bytes(ns_pad, nt_pad) <= 48 * 1024) {
    A.resident = 1;
    A.s_cap = ns_pad;
    A.t_cap = nt_pad;
  } else {
    A.resident = 0;
    A.s_cap = 512;  // 2 x 8 KiB sphere records
    A.t_cap = 256;  // 16 KiB triangle records + 1 KiB numerators
  }
  const size_t lds = lds_bytes(A.s_cap, A.t_cap);
  // only 1, 2 and 4 are instantiated; anything else keeps the default (the grid is sized from the P that runs)
  static const int forced_p = [] {
    const char *e = getenv("MRT_PIXELS_PER_LANE");
    const int x = e ? atoi(e) : 0;
    return (x == 1 || x == 2 || x == 4) ? x : 0;
  }();
  const int P = forced_p ? forced_p : 1;
  dim3 grid((A.W + 32 * P - 1) / (32 * P), (A.local_rows + 7) / 8);
  if (grid.x == 0 || grid.y == 0) return hipSuccess;
  switch (P) {
  case 1: hipLaunchKernelGGL(render_kernel<1>, grid, dim3(256), lds, stream, A); break;
  case 2: hipLaunchKernelGGL(render_kernel<2>, grid, dim3(256), lds, stream, A); break;
  default: hipLaunchKernelGGL(render_kernel<4>, grid, dim3(256), lds, stream, A); break;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// batched single-primitive intersection (the reference's hit_sphere / hit_triangle test kernels)
// ------------------------------------------------------------------------------------------

// rt::Interval::contains in its four modes (core/interval.cuh:25-42); mode bit 0: lower bound open, bit 1: upper open
__device__ __forceinline__ bool contains_mode(float lo, float hi, float f, int mode)
{
  return ((mode & 1) ? (lo < f) : (lo <= f)) && ((mode & 2) ? (f < hi) : (f <= hi));
}
// sphere_roots (mrt_device.h) with the interval's mode
__device__ __forceinline__ bool sphere_roots_mode(float a, float b, float disc, float lo, float hi, int mode, float &t_out)
{
  if (disc > 0.0f) {
    float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (contains_mode(lo, hi, t, mode)) { t_out = t; return true; }
    t = (-b + sq) / a;
    if (contains_mode(lo, hi, t, mode)) { t_out = t; return true; }
  } else if (disc == 0.0f) {
    float t = (-b) / a;
    if (contains_mode(lo, hi, t, mode)) { t_out = t; return true; }
  }
  return false;
}

__global__ __launch_bounds__(256) void intersect_kernel(int prim_kind, const float *__restrict__ prim,
                                                        const float *__restrict__ rays, float tmin, float tmax, int mode,
                                                        int64_t n, uint8_t *hit, float *t_out, float *point,
                                                        float *normal)
{
  int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float ox = rays[6 * i], oy = rays[6 * i + 1], oz = rays[6 * i + 2];
  const float dx = rays[6 * i + 3], dy = rays[6 * i + 4], dz = rays[6 * i + 5];
  bool ok = false;
  float t = 0.0f, px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0;
  if (prim_kind == 0) { // Sphere::hit, core/sphere.cuh:32-73
    const float cx = prim[0], cy = prim[1], cz = prim[2], r = prim[3];
    float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
    float a = dot3(dx, dy, dz, dx, dy, dz);
    float b = dot3(ocx, ocy, ocz, dx, dy, dz);
    float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
    float disc = b * b - a * c;
    ok = sphere_roots_mode(a, b, disc, tmin, tmax, mode, t);
    if (ok) {
      px = ox + t * dx; py = oy + t * dy; pz = oz + t * dz;
      nx = px - cx; ny = py - cy; nz = pz - cz;
      normalize3(nx, ny, nz);
    }
  } else if (prim_kind == 2) { // Triangle::hit, core/triangle.cuh:77-103
    float rdn = dot3(dx, dy, dz, prim[0], prim[1], prim[2]);
    if (rdn != 0.0f) {
      float tt = (prim[3] - dot3(ox, oy, oz, prim[0], prim[1], prim[2])) / rdn;
      float qx = ox + tt * dx, qy = oy + tt * dy, qz = oz + tt * dz;
      bool inside = (dot3(prim[4], prim[5], prim[6], qx, qy, qz) - prim[7]) > -1e-6f &&
                    (dot3(prim[8], prim[9], prim[10], qx, qy, qz) - prim[11]) > -1e-6f &&
                    (dot3(prim[12], prim[13], prim[14], qx, qy, qz) - prim[15]) > -1e-6f;
      if (inside && contains_mode(tmin, tmax, tt, mode)) {
        ok = true; t = tt; px = qx; py = qy; pz = qz;
        const bool front = rdn < 0;
        nx = front ? prim[0] : -prim[0]; ny = front ? prim[1] : -prim[1]; nz = front ? prim[2] : -prim[2];
      }
    }
  } // Plane::hit never hits (core/plane.cuh:25-29)
  hit[i] = ok ? 1 : 0;
  t_out[i] = t;
  point[3 * i] = px; point[3 * i + 1] = py; point[3 * i + 2] = pz;
  normal[3 * i] = nx; normal[3 * i + 1] = ny; normal[3 * i + 2] = nz;
}

hipError_t launch_intersect(int prim_kind, const float *prim_dev, const float *rays_dev, float tmin, float tmax, int mode,
                            int64_t n, uint8_t *hit_dev, float *t_dev, float *point_dev, float *normal_dev,
                            hipStream_t stream)
{
  if (n <= 0) return hipSuccess;
  const unsigned blocks = unsigned((n + 255) / 256);
  hipLaunchKernelGGL(intersect_kernel, dim3(blocks), dim3(256), 0, stream, prim_kind, prim_dev, rays_dev, tmin, tmax,
                     mode, n, hit_dev, t_dev, point_dev, normal_dev);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// strip scatter on rank 0 after the gather: every strip of every sending rank is three contiguous blocks (its rows
// of the rgb, id and t planes) in that rank's packed buffer and three contiguous blocks in the frame buffers.
// grid = (chunks, strips); a workgroup copies its share of its strip's three blocks, 16 bytes per thread and step
// where source, destination and size allow it.
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void copy_block(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t dwords)
{
  const size_t tid = size_t(blockIdx.x) * blockDim.x + threadIdx.x, nthr = size_t(gridDim.x) * blockDim.x;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
    const size_t n4 = dwords >> 2;
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
    uint4 *d4 = reinterpret_cast<uint4 *>(dst);
    for (size_t i = tid; i < n4; i += nthr) d4[i] = s4[i];
    for (size_t i = (n4 << 2) + tid; i < dwords; i += nthr) dst[i] = src[i];
  } else {
    for (size_t i = tid; i < dwords; i += nthr) dst[i] = src[i];
  }
}

__global__ __launch_bounds__(256) void scatter_kernel(ScatterArgs S)
{
  const int strip = blockIdx.y;
  const int r = strip % S.nranks;
  if (r >= S.skip_rank && r < S.skip_rank + S.skip_span) return; // rendered in place
  const int row0 = strip * S.strip_rows;
  const int rows = row0 + S.strip_rows <= S.H ? S.strip_rows : S.H - row0;
  const size_t lrow0 = size_t(strip / S.nranks) * S.strip_rows; // the strip's first row in its rank's buffer
  const size_t W = size_t(S.W), rr = size_t(S.rank_rows[r]);
  const uint32_t *base = S.staging + S.rank_off[r];
  copy_block(base + lrow0 * W * 3, S.rgb + size_t(row0) * W * 3, size_t(rows) * W * 3);
  // the planes a packed buffer holds follow each other: rgb | id (if gathered) | t (if gathered)
  if (S.id && S.has_id) copy_block(base + rr * W * 3 + lrow0 * W, S.id + size_t(row0) * W, size_t(rows) * W);
  if (S.t && S.has_t) copy_block(base + rr * W * (S.has_id ? 4 : 3) + lrow0 * W, S.t + size_t(row0) * W, size_t(rows) * W);
}

hipError_t launch_scatter(const ScatterArgs &S, hipStream_t stream)
{
  const int nstrips = (S.H + S.strip_rows - 1) / S.strip_rows;
  if (nstrips <= 0 || S.W <= 0) return hipSuccess;
  // enough workgroups per strip to keep about 2048 in flight, at least 4 KB of rgb each
  const size_t strip_dwords = size_t(S.strip_rows) * S.W * 3;
  unsigned chunks = unsigned((2048 + nstrips - 1) / nstrips);
  const unsigned most = unsigned((strip_dwords + 1023) / 1024);
  if (chunks > most) chunks = most;
  if (chunks < 1) chunks = 1;
  hipLaunchKernelGGL(scatter_kernel, dim3(chunks, nstrips), dim3(256), 0, stream, S);
  return hipGetLastError();
}

// The render kernels spread their ray counters over `slots` copies of 8 (same-address atomics serialise); this
// adds the copies up into the caller's 8 counters.  One workgroup: thread t owns copy t, t + 256, ...
__global__ __launch_bounds__(256) void stats_reduce_kernel(const unsigned long long *__restrict__ slots, int nslots,
                                                           unsigned long long *__restrict__ out8)
{
  __shared__ unsigned long long part[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = threadIdx.x; s < nslots; s += 256)
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += slots[size_t(s) * 8 + k];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    unsigned long long x = acc[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) part[wave][k] = x;
  }
  __syncthreads();
  if (threadIdx.x < 8) out8[threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}

hipError_t launch_stats_reduce(const unsigned long long *slots, int nslots, unsigned long long *out8, hipStream_t stream)
{
  hipLaunchKernelGGL(stats_reduce_kernel, dim3(1), dim3(256), 0, stream, slots, nslots, out8);
  return hipGetLastError();
}

} // namespace mrt
