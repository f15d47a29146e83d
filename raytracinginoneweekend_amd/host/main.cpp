// Command-line driver: main of raytracer.nim (:113-138) with its constants as defaults.
//   rtow_raytracer [--width W] [--ratio R] [--spp N] [--depth D] [--scene random|mix|redblue]
//                  [--half-extent E] [--out file.ppm] [--device K | --devices K0,K1,...] [--dump-scene]
//                  [--probe S T]   getRay(camera, S, T), then hit, rayColor and scatter of that ray: one line of numbers
//                  [--probe-one J] with --probe: also hit(Sphere) and hit(Hittable) of list entry J alone (two more lines)
//                  [--corrupt-kind J]  give list entry J a HittableKind the reference does not have (the upload must fail)
// --devices spreads the frame over several GPUs (an index may repeat); the image does not change.
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>
#include "raytracer.hpp"

using namespace rtowhost;

int main(int argc, char** argv) {
  int width = 1200, spp = 500, depth = 50, device = 0, halfExtent = 11;
  double ratio = 3.0 / 2.0;
  std::string out = "/tmp/test_small.ppm", scene = "random";
  bool dump = false, probe = false;
  int probeOne = -1, corruptKind = -1;
  double probeS = 0.5, probeT = 0.5;
  std::vector<int> devices;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
    if (a == "--width") width = std::atoi(val());
    else if (a == "--ratio") ratio = std::atof(val());
    else if (a == "--spp") spp = std::atoi(val());
    else if (a == "--depth") depth = std::atoi(val());
    else if (a == "--scene") scene = val();
    else if (a == "--half-extent") halfExtent = std::atoi(val());
    else if (a == "--out") out = val();
    else if (a == "--device") device = std::atoi(val());
    else if (a == "--devices") {
      for (const char* p = val(); *p;) {
        devices.push_back(std::atoi(p));
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
    }
    else if (a == "--dump-scene") dump = true;
    else if (a == "--probe") { probe = true; probeS = std::atof(val()); probeT = std::atof(val()); }
    else if (a == "--probe-one") probeOne = std::atoi(val());
    else if (a == "--corrupt-kind") corruptKind = std::atoi(val());
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  try {
    const Image img{width, (int)(width / ratio)};
    HittablesList world = scene == "mix" ? mixOfSpheres() : scene == "redblue" ? sceneRedBlue() : randomScene(halfExtent);
    if (corruptKind >= 0 && corruptKind < world.len()) world[corruptKind].kind = (HittableKind)1;
    if (dump) {  // one line per sphere, full precision: lets a test compare the scene with the oracle's
      for (const Hittable& h : world) {
        const Sphere& s = h.hSphere;
        const Color a = s.mat.kind == mkLambertian ? s.mat.mLambertian.albedo : s.mat.kind == mkMetal ? s.mat.mMetal.albedo : color(0, 0, 0);
        std::printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", (int)s.mat.kind, s.center[0], s.center[1],
                    s.center[2], s.radius, a[0], a[1], a[2], s.mat.kind == mkMetal ? s.mat.mMetal.fuzz : 0.0,
                    s.mat.kind == mkDielectric ? s.mat.mDielectric.ir : 0.0);
      }
      return 0;
    }
    const Camera cam = scene == "random" ? bookCamera(ratio)
                       : scene == "mix"  ? initCamera(point(3, 3, 2), point(0, 0, -1), vec3(0, 1, 0), 20, ratio, 2.0, std::sqrt(27.0))
                                         : initCamera(point(0, 0, 0), point(0, 0, -1), vec3(0, 1, 0), 90, ratio, 0.0, 1.0);
    if (probe) {  // the reference's per-ray procs through the C++ host: getRay, hit, rayColor, scatter
      Device dev(device);
      dev.setWorld(world);
      dev.setCamera(cam);
      const Ray r = getRay(dev, probeS, probeT);
      HitRecord rec;
      const bool h = hit(world, dev, r, 0.001, INFINITY, rec);
      const Color c = rayColor(r, dev, depth);
      Color att = color(0, 0, 0);
      Ray sc;
      const bool s = h && scatter(rec.mat, dev, r, rec, att, sc);
      std::printf("%.17g %.17g %.17g %.17g %.17g %.17g %d %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g %.17g %.17g "
                  "%d %.17g %.17g %.17g %.17g %.17g %.17g\n",
                  r.orig[0], r.orig[1], r.orig[2], r.dir[0], r.dir[1], r.dir[2], (int)h, rec.index, rec.t, rec.p[0], rec.p[1],
                  rec.p[2], rec.normal[0], rec.normal[1], rec.normal[2], (int)rec.frontFace, c[0], c[1], c[2],
                  (int)s, att[0], att[1], att[2], sc.dir[0], sc.dir[1], sc.dir[2]);
      if (probeOne >= 0 && probeOne < world.len()) {   // the one-object overloads of hit, against list entry probeOne alone
        for (int form = 0; form < 2; ++form) {
          HitRecord one;
          const bool h1 = form == 0 ? hit(world[probeOne].hSphere, dev, r, 0.001, INFINITY, one)
                                    : hit(world[probeOne], dev, r, 0.001, INFINITY, one);
          std::printf("%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d\n", (int)h1, one.t, one.p[0], one.p[1], one.p[2],
                      one.normal[0], one.normal[1], one.normal[2], (int)one.frontFace);
        }
      }
      return 0;
    }
    if (!devices.empty()) {
      uint64_t segments = 0;
      renderMulti(img, out, world, cam, spp, depth, devices, kRenderSeed, &segments);
      std::fprintf(stderr, "%dx%d spp %d depth %d, %d spheres on %zu contexts: %llu ray segments -> %s\n", img.width, img.height,
                   spp, depth, world.len(), devices.size(), (unsigned long long)segments, out.c_str());
      return 0;
    }
    rtow_stats st{};
    render(img, out, world, cam, spp, depth, kRenderSeed, device, &st);
    std::fprintf(stderr, "%dx%d spp %d depth %d, %d spheres: %llu ray segments, trace %.3f ms (%.1f Mrays/s) -> %s\n", img.width,
                 img.height, spp, depth, world.len(), (unsigned long long)st.ray_segments, st.trace_ms,
                 st.ray_segments / (st.trace_ms * 1e3), out.c_str());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
