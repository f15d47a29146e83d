// Host-side value types with the reference's names (basetypes.nim), binary64 like Nim's float.
// They exist to build scenes and cameras exactly as the reference does; the per-ray use of these
// operations runs on the GPU.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>

namespace rtowhost {

struct Image { int width = 0, height = 0; };  // basetypes.nim:4-5

struct Vec3 {  // basetypes.nim:7-8
  std::array<double, 3> arr{0, 0, 0};
  double& operator[](int i) { return arr[i]; }
  double operator[](int i) const { return arr[i]; }
  double x() const { return arr[0]; }
  double y() const { return arr[1]; }
  double z() const { return arr[2]; }
};
using Color = Vec3;  // distinct types in Nim (basetypes.nim:10-11); aliases suffice on the host
using Point = Vec3;

inline Vec3 vec3(double x, double y, double z) { return Vec3{{x, y, z}}; }
inline Color color(double r, double g, double b) { return Vec3{{r, g, b}}; }
inline Point point(double x, double y, double z) { return Vec3{{x, y, z}}; }

// makeMath (basetypes.nim:68-81)
#define RTOW_VEC_OP(op)                                                                                   \
  inline Vec3 operator op(const Vec3& v, const Vec3& u) { return vec3(v[0] op u[0], v[1] op u[1], v[2] op u[2]); } \
  inline Vec3 operator op(const Vec3& v, double s) { return vec3(v[0] op s, v[1] op s, v[2] op s); }      \
  inline Vec3 operator op(double s, const Vec3& v) { return vec3(s op v[0], s op v[1], s op v[2]); }
RTOW_VEC_OP(+)
RTOW_VEC_OP(-)
RTOW_VEC_OP(*)
RTOW_VEC_OP(/)
#undef RTOW_VEC_OP
inline Vec3 operator-(const Vec3& v) { return vec3(-v[0], -v[1], -v[2]); }

inline double dot(const Vec3& v, const Vec3& u) {  // basetypes.nim:25-27
  double r = 0.0;
  for (int i = 0; i < 3; ++i) r += v[i] * u[i];
  return r;
}
inline Vec3 cross(const Vec3& v, const Vec3& u) {  // basetypes.nim:29-32
  return vec3(v[1] * u[2] - v[2] * u[1], v[2] * u[0] - v[0] * u[2], v[0] * u[1] - v[1] * u[0]);
}
inline double length_squared(const Vec3& v) { return v.x() * v.x() + v.y() * v.y() + v.z() * v.z(); }
inline double length(const Vec3& v) { return std::sqrt(length_squared(v)); }
inline Vec3 normalize(const Vec3& v) { return v / length(v); }
inline Vec3 unitVector(const Vec3& v) { return normalize(v); }

struct Ray { Point orig; Vec3 dir; };  // basetypes.nim:13-15
inline Ray initRay(const Point& o, const Vec3& d) { return Ray{o, d}; }
inline Point at(const Ray& r, double t) { return r.orig + t * r.dir; }

// Nim std/random as the reference uses it for scene construction: xoroshiro128+ from the
// un-seeded global state (randomScene runs before randomize, raytracer.nim:122 vs :135).
class NimRand {
 public:
  uint64_t a0 = 0x69B4C98CB8530805ULL, a1 = 0xD1DD3004688D67CAULL;
  uint64_t draws = 0;
  uint64_t next() {
    const uint64_t s0 = a0;
    uint64_t s1 = a1;
    const uint64_t result = s0 + s1;
    s1 ^= s0;
    a0 = rotl(s0, 55) ^ s1 ^ (s1 << 14);
    a1 = rotl(s1, 36);
    ++draws;
    return result;
  }
  double rand(double max) {  // rand(max: float)
    const uint64_t bits = (0x3FFULL << 52) | (next() >> 12);
    double d;
    static_assert(sizeof d == sizeof bits, "binary64");
    __builtin_memcpy(&d, &bits, sizeof d);
    return (d - 1.0) * max;
  }
  double rand(double lo, double hi) { return rand(hi - lo) + lo; }  // rand(lo .. hi)
 private:
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
};

inline Vec3 randomVec(NimRand& g, double lo = 0.0, double hi = 1.0) {  // basetypes.nim:110-112
  const double x = g.rand(lo, hi), y = g.rand(lo, hi), z = g.rand(lo, hi);
  return vec3(x, y, z);
}

}  // namespace rtowhost
