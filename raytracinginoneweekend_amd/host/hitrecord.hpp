// Sphere and material descriptions with the reference's names (hitrecord.nim:4-49).
#pragma once
#include "../../include/rtow.h"
#include "basetypes.hpp"

namespace rtowhost {

enum MaterialKind { mkLambertian = RTOW_LAMBERTIAN, mkMetal = RTOW_METAL, mkDielectric = RTOW_DIELECTRIC };

struct Lambertian { Color albedo; };
struct Metal { Color albedo; double fuzz = 0; };
struct Dielectric { double ir = 1.5; };

struct Material {  // hitrecord.nim:18-22: a tagged union
  MaterialKind kind = mkLambertian;
  Lambertian mLambertian;
  Metal mMetal;
  Dielectric mDielectric;
};

inline Lambertian initLambertian(const Color& a) { return Lambertian{a}; }
inline Metal initMetal(const Color& a, double f) { return Metal{a, f}; }
inline Dielectric initDielectric(double ir) { return Dielectric{ir}; }
inline Material initMaterial(const Lambertian& m) { Material r; r.kind = mkLambertian; r.mLambertian = m; return r; }
inline Material initMaterial(const Metal& m) { Material r; r.kind = mkMetal; r.mMetal = m; return r; }
inline Material initMaterial(const Dielectric& m) { Material r; r.kind = mkDielectric; r.mDielectric = m; return r; }

struct Sphere {  // hitrecord.nim:11-14
  Point center;
  double radius = 0;
  Material mat;
};

struct HitRecord {  // hitrecord.nim:4-9
  Point p;
  Vec3 normal;
  double t = 0;
  bool frontFace = false;
  Material mat;
  int index = -1;  // position of the sphere in its list (not in the reference: a debugging aid)
  void setFaceNormal(const Ray& r, const Vec3& outward_normal) {  // hitrecord.nim:51-53
    frontFace = dot(r.dir, outward_normal) < 0;
    normal = frontFace ? outward_normal : -outward_normal;
  }
};

}  // namespace rtowhost
