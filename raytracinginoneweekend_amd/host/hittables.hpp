// The scene container with the reference's interface (hittables.nim:3-73) and the flattening to
// the structure-of-arrays form rtow_set_scene_f64 uploads.
#pragma once
#include <cassert>
#include <cstdint>
#include <vector>
#include "hitrecord.hpp"

namespace rtowhost {

enum HittableKind { htSphere };   // hittables.nim:4-5: one member; the value travels to the library, which refuses any other
struct Hittable { HittableKind kind = htSphere; Sphere hSphere; };  // hittables.nim:4-8

class HittablesList {
 public:
  explicit HittablesList(int size = 8) { data_.reserve(size < 8 ? 8 : size); }
  int len() const { return (int)data_.size(); }
  const Hittable& operator[](int idx) const { assert(idx < len()); return data_[(size_t)idx]; }
  Hittable& operator[](int idx) { assert(idx < len()); return data_[(size_t)idx]; }
  void add(const Hittable& el) { data_.push_back(el); }     // hittables.nim:59-64
  void add(const Sphere& s) { Hittable h; h.hSphere = s; add(h); }  // hittables.nim:66-69
  std::vector<Hittable>::const_iterator begin() const { return data_.begin(); }  // items, :71-73
  std::vector<Hittable>::const_iterator end() const { return data_.end(); }

  struct Flat {
    std::vector<double> cx, cy, cz, radius, albedo, fuzz, ir;   // Nim float = binary64
    std::vector<uint8_t> kind;           // MaterialKind of the sphere's material
    std::vector<uint8_t> hittableKind;   // Hittable.kind (hittables.nim:6), for rtow_set_hittables
  };
  // list order is kept: index i here is insertion index i, which decides ties (hittables.nim:84-88)
  Flat flatten() const {
    Flat f;
    for (const Hittable& h : data_) {
      const Sphere& s = h.hSphere;
      f.hittableKind.push_back((uint8_t)h.kind);
      f.cx.push_back(s.center[0]); f.cy.push_back(s.center[1]); f.cz.push_back(s.center[2]);
      f.radius.push_back(s.radius);
      f.kind.push_back((uint8_t)s.mat.kind);
      const Color a = s.mat.kind == mkLambertian ? s.mat.mLambertian.albedo
                      : s.mat.kind == mkMetal    ? s.mat.mMetal.albedo : color(1, 1, 1);
      for (int c = 0; c < 3; ++c) f.albedo.push_back(a[c]);
      f.fuzz.push_back(s.mat.kind == mkMetal ? s.mat.mMetal.fuzz : 0.0);
      f.ir.push_back(s.mat.kind == mkDielectric ? s.mat.mDielectric.ir : 1.0);
    }
    return f;
  }
 private:
  std::vector<Hittable> data_;
};

inline HittablesList initHittables(int size = 8) { return HittablesList(size); }

}  // namespace rtowhost
