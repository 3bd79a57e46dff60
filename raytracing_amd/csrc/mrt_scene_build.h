// mrt_scene_build.h -- host-side preparation of a scene at upload (included by mrt_api.cpp only): primitive
// order, cluster bounds, the conservative bound of a triangle's accepted region, shadow-ray edge functions.
// Pure host arithmetic, camera independent; nothing here touches HIP.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mrt.h"

namespace mrt {
namespace build {

// ---- primitive-side hierarchy (camera independent, built once per upload) ----
// Primitives of each type are stored in kd order (default; Morton order for ablation) of their bound centres, and every run of 64 consecutive
// primitives ("cluster") gets a bounding ball.  The culling kernels test clusters first.  Reordering is
// invisible in the results: hits are reported by object number, and the closest-hit rule
// (t, object number) lexicographic minimum does not depend on the visiting order.
inline uint32_t spread10(uint32_t v)
{
  v &= 0x3ff;
  v = (v | (v << 16)) & 0x030000ff;
  v = (v | (v << 8)) & 0x0300f00f;
  v = (v | (v << 4)) & 0x030c30c3;
  v = (v | (v << 2)) & 0x09249249;
  return v;
}

// order[i] = index of the primitive stored at position i; bound(i, out4) gives centre + radius
template <class BoundFn> inline void morton_order(size_t n, BoundFn bound, std::vector<uint32_t> &order)
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
inline void kd_split(uint32_t *idx, size_t lo, size_t hi, const float *c)
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

// MRT_LOOSE_FACTOR (ablation; 0 = keep loose primitives where the kd order puts them)
inline float loose_factor()
{
  static const float f = [] { const char *e = getenv("MRT_LOOSE_FACTOR"); return e ? (float)atof(e) : 8.0f; }();
  return f;
}

// *regular (optional): how many primitives at the head of the order are neither loose nor unbounded
template <class BoundFn> inline void kd_order(size_t n, BoundFn bound, std::vector<uint32_t> &order, size_t *regular = nullptr)
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
  // Primitives with a far larger bound than the rest (the long wedges the reference's edge planes accept for some
  // triangles: 6 395 of the 1M-triangle terrain) get clusters of their own behind the regular ones: inside a regular
  // cluster a single one of them inflates the cluster's ball until every tile opens it.  "Far larger" = more than
  // loose_factor() times the median radius.
  std::vector<uint32_t> loose;
  if (order.size() > 128 && loose_factor() > 0.0f) {
    std::vector<float> radii(order.size());
    for (size_t i = 0; i < order.size(); ++i) radii[i] = c[4 * order[i] + 3];
    std::nth_element(radii.begin(), radii.begin() + radii.size() / 2, radii.end());
    const float limit = loose_factor() * radii[radii.size() / 2];
    std::vector<uint32_t> regular;
    regular.reserve(order.size());
    for (uint32_t i : order) (c[4 * i + 3] > limit ? loose : regular).push_back(i);
    order.swap(regular);
  }
  if (order.size() > 64) kd_split(order.data(), 0, order.size(), c.data());
  if (loose.size() > 64) kd_split(loose.data(), 0, loose.size(), c.data());
  if (regular) *regular = order.size();
  order.insert(order.end(), loose.begin(), loose.end());
  order.insert(order.end(), unbounded.begin(), unbounded.end());
}

// MRT_ORDER=morton selects the Morton order (ablation); default: kd order
template <class BoundFn> inline void primitive_order(size_t n, BoundFn bound, std::vector<uint32_t> &order, size_t *regular = nullptr)
{
  static const bool morton = [] { const char *e = getenv("MRT_ORDER"); return e && std::strcmp(e, "morton") == 0; }();
  if (regular) *regular = n;
  if (morton) morton_order(n, bound, order);
  else kd_order(n, bound, order, regular);
}

// bounding ball of each run of 64 balls (streams x,y,z,r of n floats) -> 4 streams of ceil(n/64) floats
inline void cluster_bounds(const float *x, const float *y, const float *z, const float *r, size_t n, float *out)
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

// Conservative bounding sphere of the region Triangle::hit accepts (core/triangle.cuh:89-91): the set of
// points p with n_i.p - k_i > -1e-6 for the three edge planes, near the plane pn.p = kp.  Derived from the
// 16 precomputed floats only (the hit test never sees vertices, and for degenerate triangles -- e.g. a
// plane through the world origin, SURVEY section 0.6 -- the accepted region is NOT the geometric triangle).
// Unbounded, empty or ill-conditioned regions get radius +inf, i.e. are never culled.
inline void tri_bound(const float q[16], float out[4])
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

inline void shadow_functions(const float q[16], int kind, const float lv[3], float out[SH_STRIDE])
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


} // namespace build
} // namespace mrt
