// Driver with the reference's interface (raytracer.nim): render, the scene builders, the PPM
// writer.  render() keeps the reference's signature; its j/i/s loops (raytracer.nim:36-45) are
// one call into librtow_hip.so.
#pragma once
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/rtow.h"
#include "camera.hpp"
#include "hittables.hpp"

namespace rtowhost {

constexpr uint64_t kRenderSeed = 0xE7;  // raytracer.nim:135

inline void check(int status, const char* what) {
  if (status != RTOW_OK) throw std::runtime_error(std::string(what) + ": " + rtow_last_error());
}

// The P3 container of raytracer.nim:35,28: header, one "r g b" line per pixel, top row first.
inline void writePPM(const std::string& path, int w, int h, const std::vector<uint8_t>& rgb) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::fprintf(f, "P3\n%d %d\n255\n", w, h);
  for (size_t p = 0; p < (size_t)w * (size_t)h; ++p)
    std::fprintf(f, "%d %d %d\n", rgb[3 * p], rgb[3 * p + 1], rgb[3 * p + 2]);
  std::fclose(f);
}

// Upload a list as the reference holds it: Hittable.kind per entry beside the sphere arrays (the library has, like
// hit*(h: Hittable) of hittables.nim:75-77, the one branch htSphere and refuses anything else).
inline void upload(rtow_ctx* ctx, const HittablesList& world) {
  const HittablesList::Flat s = world.flatten();
  check(rtow_set_hittables(ctx, world.len(), s.hittableKind.data(), s.cx.data(), s.cy.data(), s.cz.data(), s.radius.data(),
                           s.kind.data(), s.albedo.data(), s.fuzz.data(), s.ir.data()), "rtow_set_hittables");
}

// render*(img, f, world, camera, samplesPerPixel, maxDepth)  (raytracer.nim:30-46)
inline std::vector<uint8_t> render(const Image& img, const std::string& f, const HittablesList& world,
                                   const Camera& camera, int samplesPerPixel, int maxDepth,
                                   uint64_t seed = kRenderSeed, int device = 0, rtow_stats* stats = nullptr) {
  rtow_ctx* ctx = nullptr;
  check(rtow_create(device, &ctx), "rtow_create");
  std::vector<uint8_t> rgb((size_t)img.width * (size_t)img.height * 3);
  try {
    upload(ctx, world);
    const rtow_camera_f64 cam = camera.flat();
    check(rtow_set_camera_f64(ctx, &cam), "rtow_set_camera_f64");
    check(rtow_render(ctx, img.width, img.height, samplesPerPixel, maxDepth, seed, nullptr, img.height,
                      rgb.data(), nullptr, nullptr), "rtow_render");
    if (stats) check(rtow_get_stats(ctx, stats), "rtow_get_stats");
    if (!f.empty()) {  // the P3 text is composed on the device; writePPM() above is its host-side equivalent
      uint64_t n = 0;
      check(rtow_format_ppm(ctx, 0, nullptr, 0, &n), "rtow_format_ppm");
      std::vector<char> text((size_t)n);
      check(rtow_format_ppm(ctx, 0, text.data(), n, &n), "rtow_format_ppm");
      FILE* out = std::fopen(f.c_str(), "wb");
      if (!out) throw std::runtime_error("cannot open " + f);
      const size_t put = std::fwrite(text.data(), 1, text.size(), out);
      if (std::fclose(out) != 0 || put != text.size()) throw std::runtime_error("cannot write " + f);
    }
  } catch (...) {
    rtow_destroy(ctx);
    throw;
  }
  rtow_destroy(ctx);
  return rgb;
}

// The same frame spread over several GPUs of this process (devices[k] = HIP device index; an index
// may repeat): one context per entry, rows dealt in interleaved 4-row tiles, image assembled on the
// host.  The file is byte-identical to what render() writes.
inline std::vector<uint8_t> renderMulti(const Image& img, const std::string& f, const HittablesList& world,
                                        const Camera& camera, int samplesPerPixel, int maxDepth,
                                        const std::vector<int>& devices, uint64_t seed = kRenderSeed,
                                        uint64_t* raySegments = nullptr) {
  std::vector<rtow_ctx*> ctxs;
  std::vector<uint8_t> rgb((size_t)img.width * (size_t)img.height * 3);
  auto destroyAll = [&]() { for (rtow_ctx* c : ctxs) rtow_destroy(c); };
  try {
    const rtow_camera_f64 cam = camera.flat();
    for (int device : devices) {
      rtow_ctx* ctx = nullptr;
      check(rtow_create(device, &ctx), "rtow_create");
      ctxs.push_back(ctx);
      upload(ctx, world);
      check(rtow_set_camera_f64(ctx, &cam), "rtow_set_camera_f64");
    }
    check(rtow_render_multi(ctxs.data(), (int)ctxs.size(), img.width, img.height, samplesPerPixel, maxDepth, seed, 4,
                            rgb.data(), raySegments), "rtow_render_multi");
  } catch (...) {
    destroyAll();
    throw;
  }
  destroyAll();
  if (!f.empty()) writePPM(f, img.width, img.height, rgb);
  return rgb;
}

// A context that holds one world and one camera for the per-ray procs below (they would otherwise upload
// the scene for every ray).  The reference's procs take the world / camera by value; these take a Device.
class Device {
 public:
  explicit Device(int device = 0) { check(rtow_create(device, &ctx_), "rtow_create"); }
  ~Device() { rtow_destroy(ctx_); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  void setWorld(const HittablesList& world) { upload(ctx_, world); }
  void setCamera(const Camera& camera) {
    const rtow_camera_f64 cam = camera.flat();
    check(rtow_set_camera_f64(ctx_, &cam), "rtow_set_camera_f64");
  }
  rtow_ctx* ctx() const { return ctx_; }
 private:
  rtow_ctx* ctx_ = nullptr;
};

// rayColor*(r, world, depth)  (raytracer.nim:5-20): the world is the one `dev` holds; scattering draws from the
// stream of path (pixel 0, sample 0) of `seed` instead of the reference's global generator
inline Color rayColor(const Ray& r, Device& dev, int depth, uint64_t seed = kRenderSeed) {
  double c[3];
  check(rtow_trace_rays(dev.ctx(), 1, r.orig.arr.data(), r.dir.arr.data(), 0.001, INFINITY, depth, seed, c, nullptr, nullptr), "rtow_trace_rays");
  return color(c[0], c[1], c[2]);
}

// hit*(h, r, t_min, t_max, rec)  (hittables.nim:79-88): the search runs on the GPU over the list `dev` holds (which
// must be `h`); the record is completed as Sphere.hit does (hitrecord.nim:73-77)
inline bool hit(const HittablesList& h, Device& dev, const Ray& r, double t_min, double t_max, HitRecord& rec) {
  int32_t id = -1;
  double t = 0;
  check(rtow_trace_rays(dev.ctx(), 1, r.orig.arr.data(), r.dir.arr.data(), t_min, t_max, 0, kRenderSeed, nullptr, &id, &t), "rtow_trace_rays");
  if (id < 0) return false;
  const Sphere& s = h[id].hSphere;
  rec.t = t;
  rec.p = at(r, t);
  rec.setFaceNormal(r, (rec.p - s.center) / s.radius);
  rec.mat = s.mat;
  rec.index = id;
  return true;
}

// hit*(s: Sphere, r, t_min, t_max, rec)  (hitrecord.nim:55-79): one sphere is a one-entry list.  `dev` is given that
// list (whatever world it held before is replaced).
inline bool hit(const Sphere& s, Device& dev, const Ray& r, double t_min, double t_max, HitRecord& rec) {
  HittablesList one = initHittables(0);
  one.add(s);
  dev.setWorld(one);
  return hit(one, dev, r, t_min, t_max, rec);
}

// hit*(h: Hittable, r, t_min, t_max, rec)  (hittables.nim:75-77): case h.kind of htSphere: h.hSphere.hit(...)
inline bool hit(const Hittable& h, Device& dev, const Ray& r, double t_min, double t_max, HitRecord& rec) {
  switch (h.kind) {
    case htSphere: return hit(h.hSphere, dev, r, t_min, t_max, rec);
  }
  throw std::runtime_error("hit: Hittable of unknown kind " + std::to_string((int)h.kind));
}

// scatter*(m, r_in, rec, attenuation, scattered)  (hitrecord.nim:127-132); draws from the stream of path
// (pixel 0, sample 0) of `seed` instead of the reference's global generator
inline bool scatter(const Material& m, Device& dev, const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered,
                    uint64_t seed = kRenderSeed) {
  const uint8_t kind = (uint8_t)m.kind, front = rec.frontFace ? 1 : 0;
  const Color a = m.kind == mkLambertian ? m.mLambertian.albedo : m.kind == mkMetal ? m.mMetal.albedo : color(1, 1, 1);
  const double fuzz = m.kind == mkMetal ? m.mMetal.fuzz : 0.0, ir = m.kind == mkDielectric ? m.mDielectric.ir : 1.0;
  uint8_t ok = 0;
  scattered.orig = rec.p;
  check(rtow_scatter_rays(dev.ctx(), 1, r_in.dir.arr.data(), rec.normal.arr.data(), &front, &kind, a.arr.data(), &fuzz, &ir, seed,
                          attenuation.arr.data(), scattered.dir.arr.data(), &ok), "rtow_scatter_rays");
  return ok != 0;
}

// getRay*(c, s, t)  (camera.nim:34-39) for the camera `dev` holds
inline Ray getRay(Device& dev, double s, double t, uint64_t seed = kRenderSeed) {
  Ray r;
  check(rtow_get_rays(dev.ctx(), 1, &s, &t, seed, r.orig.arr.data(), r.dir.arr.data()), "rtow_get_rays");
  return r;
}

inline HittablesList sceneRedBlue() {  // raytracer.nim:48-59
  HittablesList result = initHittables(0);
  const double R = std::cos(3.14159265358979323846 / 4.0);
  result.add(Sphere{point(-R, 0, -1), R, initMaterial(initLambertian(color(0, 0, 1)))});
  result.add(Sphere{point(R, 0, -1), R, initMaterial(initLambertian(color(1, 0, 0)))});
  return result;
}

inline HittablesList mixOfSpheres() {  // raytracer.nim:61-73 (fourth sphere: negative radius)
  HittablesList result = initHittables(0);
  const Material matLeft = initMaterial(initDielectric(1.5));
  result.add(Sphere{point(0, -100.5, -1), 100, initMaterial(initLambertian(color(0.8, 0.8, 0.0)))});
  result.add(Sphere{point(0, 0, -1), 0.5, initMaterial(initLambertian(color(0.1, 0.2, 0.5)))});
  result.add(Sphere{point(-1.0, 0, -1), 0.5, matLeft});
  result.add(Sphere{point(-1.0, 0, -1), -0.4, matLeft});
  result.add(Sphere{point(1.0, 0, -1), 0.5, initMaterial(initMetal(color(0.8, 0.6, 0.2), 1.0))});
  return result;
}

// randomScene (raytracer.nim:75-111); halfExtent 11 and thresholds 0.8 / 0.95 are the reference's
inline HittablesList randomScene(int halfExtent = 11, double lambertBelow = 0.8, double metalBelow = 0.95,
                                 NimRand* rng = nullptr) {
  NimRand local;
  NimRand& g = rng ? *rng : local;
  HittablesList result = initHittables(0);
  result.add(Sphere{point(0, -1000, 0), 1000, initMaterial(initLambertian(color(0.5, 0.5, 0.5)))});
  for (int a = -halfExtent; a < halfExtent; ++a)
    for (int b = -halfExtent; b < halfExtent; ++b) {
      const double chooseMat = g.rand(1.0);
      const double cx = (double)a + 0.9 * g.rand(1.0);
      const double cz = (double)b + 0.9 * g.rand(1.0);
      const Point center = point(cx, 0.2, cz);
      if (length(center - point(4, 0.2, 0)) > 0.9) {
        Material m;
        if (chooseMat < lambertBelow) {
          const Vec3 first = randomVec(g);
          const Vec3 second = randomVec(g);
          m = initMaterial(initLambertian(first * second));
        } else if (chooseMat < metalBelow) {
          const Vec3 albedo = randomVec(g, 0.5, 1.0);
          const double fuzz = g.rand(0.0, 0.5);
          m = initMaterial(initMetal(albedo, fuzz));
        } else {
          m = initMaterial(initDielectric(1.5));
        }
        result.add(Sphere{center, 0.2, m});
      }
    }
  result.add(Sphere{point(0, 1, 0), 1.0, initMaterial(initDielectric(1.5))});
  result.add(Sphere{point(-4, 1, 0), 1.0, initMaterial(initLambertian(color(0.4, 0.2, 0.1)))});
  result.add(Sphere{point(4, 1, 0), 1.0, initMaterial(initMetal(color(0.7, 0.6, 0.5), 0.0))});
  return result;
}

inline Camera bookCamera(double aspectRatio) {  // raytracer.nim:125-132
  return initCamera(point(13, 2, 3), point(0, 0, 0), vec3(0, 1, 0), 20, aspectRatio, 0.1, 10.0);
}

}  // namespace rtowhost
