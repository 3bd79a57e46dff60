// math/vec.hpp -- the 3-vector arithmetic the host layer uses in place of Eigen.
//
// The reference does its vector math with Eigen (deps/eigen, un-vendored, no pinned
// version).  This header restates the evaluation order of Eigen's non-vectorised
// fixed-size path, which is the numerical contract shared with the CPU oracle and
// the gfx950 kernels (DESIGN.md "Arithmetic contract"):
//   dot(a,b)      = a0*b0 + (a1*b1 + a2*b2)
//   cross(a,b)    = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
//   normalized(v) = z = dot(v,v); z > 0 ? v / sqrt(z) : v   (true division)
//   a + s*b etc. evaluate component-wise, left to right.
// Build host code with -ffp-contract=off.
#pragma once

#include <cmath>

namespace rt {

struct Vec2f {
  float x = 0.0f, y = 0.0f;
  Vec2f() = default;
  Vec2f(float x_, float y_) : x(x_), y(y_) {}
};

inline Vec2f operator+(Vec2f a, Vec2f b) { return {a.x + b.x, a.y + b.y}; }

struct Vec3f {
  float x = 0.0f, y = 0.0f, z = 0.0f;
  Vec3f() = default;
  Vec3f(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}

  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }

  float dot(const Vec3f &b) const { return x * b.x + (y * b.y + z * b.z); }
  Vec3f cross(const Vec3f &b) const { return {y * b.z - z * b.y, z * b.x - x * b.z, x * b.y - y * b.x}; }
  float squaredNorm() const { return dot(*this); }
  Vec3f normalized() const
  {
    float n2 = squaredNorm();
    if (n2 > 0.0f) {
      float n = std::sqrt(n2);
      return {x / n, y / n, z / n};
    }
    return *this;
  }
  Vec3f &operator+=(const Vec3f &b) { x += b.x; y += b.y; z += b.z; return *this; }
  Vec3f &operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
  Vec3f &operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
};

inline Vec3f operator+(const Vec3f &a, const Vec3f &b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3f operator-(const Vec3f &a, const Vec3f &b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3f operator-(const Vec3f &a) { return {-a.x, -a.y, -a.z}; }
inline Vec3f operator*(float s, const Vec3f &a) { return {s * a.x, s * a.y, s * a.z}; }
inline Vec3f operator*(const Vec3f &a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline Vec3f operator/(const Vec3f &a, float s) { return {a.x / s, a.y / s, a.z / s}; }

// row-major 3x3, only what Transform needs
struct Mat3f {
  float m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  static Mat3f identity() { return Mat3f{}; }
  Vec3f operator*(const Vec3f &v) const
  {
    // Eigen's lazy 3x3 * 3x1 product: each coefficient is a length-3 redux
    return {m[0][0] * v.x + (m[0][1] * v.y + m[0][2] * v.z), m[1][0] * v.x + (m[1][1] * v.y + m[1][2] * v.z),
            m[2][0] * v.x + (m[2][1] * v.y + m[2][2] * v.z)};
  }
};

} // namespace rt
