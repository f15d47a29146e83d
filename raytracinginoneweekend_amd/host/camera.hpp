// Camera with the reference's interface (camera.nim): initCamera runs once on the host in
// binary64 (camera.nim:12-31); getRay (camera.nim:34-39) runs on the GPU.
#pragma once
#include "../../include/rtow.h"
#include "basetypes.hpp"

namespace rtowhost {

struct Camera {  // camera.nim:4-10
  Point origin, lowerLeftCorner;
  Vec3 horizontal, vertical;
  double lensRadius = 0;
  Vec3 u, v, w;

  rtow_camera_f64 flat() const {  // the eight fields as computed, for rtow_set_camera_f64
    rtow_camera_f64 c;
    for (int i = 0; i < 3; ++i) {
      c.origin[i] = origin[i]; c.lower_left[i] = lowerLeftCorner[i];
      c.horizontal[i] = horizontal[i]; c.vertical[i] = vertical[i];
      c.u[i] = u[i]; c.v[i] = v[i]; c.w[i] = w[i];
    }
    c.lens_radius = lensRadius;
    return c;
  }
};

inline Camera initCamera(const Point& lookFrom, const Point& lookAt, const Vec3& vup, double vfov,
                         double aspectRatio, double aperture, double focusDist) {
  const double theta = vfov * (3.14159265358979323846 / 180.0);  // degToRad
  const double h = std::tan(theta / 2.0);
  const double viewHeight = 2.0 * h;
  const double viewWidth = aspectRatio * viewHeight;
  Camera c;
  c.w = unitVector(lookFrom - lookAt);
  c.u = unitVector(cross(vup, c.w));
  c.v = cross(c.w, c.u);
  c.origin = lookFrom;
  c.horizontal = focusDist * viewWidth * c.u;
  c.vertical = focusDist * viewHeight * c.v;
  c.lowerLeftCorner = c.origin - c.horizontal / 2.0 - c.vertical / 2.0 - focusDist * c.w;
  c.lensRadius = aperture / 2.0;
  return c;
}

}  // namespace rtowhost
