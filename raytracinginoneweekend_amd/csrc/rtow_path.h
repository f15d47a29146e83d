// Per-path arithmetic of the renderer: random streams, camera rays, candidate roots, material
// scattering, radiance accumulation.  Everything here is a pure function of its arguments and is
// written so that each rounding step is explicit (one IEEE binary32 operation per operator,
// fused only where fmaf is spelled out).  The translation units that include this header are
// compiled with -ffp-contract=off; the kernels add the wave-level scheduling and the sphere scan
// around these functions.
//
// Reference being accelerated (Nim, FP64): raytracer.nim:5-46, camera.nim:34-39,
// hitrecord.nim:51-132, hittables.nim:79-88, basetypes.nim:87-134.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RTOW_HD __host__ __device__ __forceinline__
#else
#define RTOW_HD inline
#endif

namespace rtow {

constexpr float kBigRadius = 100.0f;        // |r| >= this: intersect in binary64
// Radiance sums are 64-bit integers in units of 2^-40.  A sample computed in binary32 is cut to 30
// fractional bits first (it then fits 32 bits) and widened by a shift; a sample computed in binary64 - the
// sky colour of a camera ray that hits nothing - keeps all 40, so that the 8-bit value of a one-sample
// pixel is the reference's except when its channel lies within 2^-40 above a quantisation boundary.
constexpr float kFixOne = 1073741824.0f;    // 2^30: scale of a binary32 sample
constexpr int kSumShift = 10;               // 2^30 -> 2^40
constexpr double kSumOneD = 1099511627776.0;   // 2^40: scale of the sums
constexpr int kMaxSamplesPerPixel = 1 << 22;   // 2^22 * 3.5 * 2^40 < 2^64
constexpr float kRadianceClamp = 3.5f;      // keeps one binary32 sample's channel inside 32 bits
constexpr float kTMin = 0.001f;             // raytracer.nim:11, in units of |dir|
constexpr uint32_t kGolden = 0x9E3779B9u;

enum : int { kLambertian = 0, kMetal = 1, kDielectric = 2 };

struct f3 { float x, y, z; };

RTOW_HD float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

// ------------------------------------------------------------------ sqrt and reciprocal
// Correctly rounded, like sqrtf(x) and 1.0f / x, whose results they reproduce for EVERY input (the
// library's self test runs all 2^32 bit patterns through both).  The compiler's own sequences
// spend a third of their instructions on scaling subnormal and huge operands; these keep the
// hardware estimate plus the rounding correction and send the operands that need scaling - which
// this renderer only meets by accident - to the compiler's version.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTOW_COMPILER_SQRT_RCP)
static __device__ __forceinline__ float rt_sqrt(float x) {
  // the estimate flushes subnormals (of either sign) to zero and the residuals of tiny operands underflow
  if (__builtin_expect(fabsf(x) < 0x1p-96f && x != 0.0f, 0)) return sqrtf(x);
  float s = __builtin_amdgcn_sqrtf(x);                                 // within 1 ulp
  const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
  const float r_dn = fmaf(-s_dn, s, x), r_up = fmaf(-s_up, s, x);      // exact signs of x - s_dn*s and x - s_up*s
  s = r_dn <= 0.0f ? s_dn : s;
  s = r_up > 0.0f ? s_up : s;
  return s;
}
static __device__ __forceinline__ float rt_rcp(float x) {
  const float ax = fabsf(x);
  if (__builtin_expect(!(ax >= 0x1p-126f && ax <= 0x1p126f), 0)) return 1.0f / x;  // subnormal operand or result, 0, inf, NaN
  const float y = __builtin_amdgcn_rcpf(x);                                        // within 1 ulp
  return fmaf(fmaf(-x, y, 1.0f), y, y);                                            // one Newton step lands on the rounded quotient
}
#else
RTOW_HD float rt_sqrt(float x) { return sqrtf(x); }
RTOW_HD float rt_rcp(float x) { return 1.0f / x; }
#endif

// ------------------------------------------------------------------ random stream of a path
// Draw k of path (pixel, sample) is hash(key0 + k*golden) with key1 folded in half way:
// a pure function of (seed, pixel, sample, k).  It stands in for the reference's global
// std/random stream (raytracer.nim:41-42, basetypes.nim:112,132, hitrecord.nim:119).
RTOW_HD uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

struct Stream {
  uint32_t x0;    // key0 + k*golden for the next draw
  uint32_t key1;
};

RTOW_HD Stream stream_begin(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel, uint32_t sample) {
  Stream g;
  g.x0 = mix32(mix32(pixel ^ seed_lo) + sample);
  g.key1 = mix32(mix32(sample ^ seed_hi ^ 0xA511E9B3U) + pixel);
  return g;
}

RTOW_HD uint32_t draw_u32(Stream& g) {
  uint32_t x = g.x0;
  g.x0 += kGolden;
  x ^= x >> 16; x *= 0x7feb352dU;
  x ^= x >> 15; x ^= g.key1; x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// uniform in [0,1), 24 random bits: rand(1.0)
RTOW_HD float draw(Stream& g) { return (float)(draw_u32(g) >> 8) * (1.0f / 16777216.0f); }
// rand(-1.0 .. 1.0) = rand(2.0) - 1.0 = (k - 2^23) / 2^23 for the 24-bit integer k: every step is
// exact in binary32, so one fused operation on k gives the same bits as 2 * draw - 1
RTOW_HD float draw_pm1(Stream& g) { return fmaf((float)(draw_u32(g) >> 8), 1.0f / 8388608.0f, -1.0f); }

// randomInUnitSphere (basetypes.nim:114-118): x, y, z drawn in order, rejected while |p|^2 >= 1
#ifdef RTOW_UNIT_BALL_TWO_TRIPS
// Experiment: two trips of the rejection loop per pass, evaluated side by side (independent
// instruction chains); the first accepted one wins and the stream advances exactly as far as the
// sequential loop would have taken it, so the result is the same bit for bit.
RTOW_HD f3 unit_ball(Stream& g, float& l2) {
  for (;;) {
    Stream a = g;
    f3 p, q;
    p.x = draw_pm1(a); p.y = draw_pm1(a); p.z = draw_pm1(a);
    Stream b = a;
    q.x = draw_pm1(b); q.y = draw_pm1(b); q.z = draw_pm1(b);
    const float lp = dot3(p, p), lq = dot3(q, q);
    if (lp < 1.0f) { g = a; l2 = lp; return p; }
    g = b;
    if (lq < 1.0f) { l2 = lq; return q; }
  }
}
#else
RTOW_HD f3 unit_ball(Stream& g, float& l2) {
  f3 p;
  do {
    p.x = draw_pm1(g); p.y = draw_pm1(g); p.z = draw_pm1(g);
    l2 = dot3(p, p);
  } while (l2 >= 1.0f);
  return p;
}
#endif

// ------------------------------------------------------------------ camera (camera.nim:4-10,34-39)
struct CameraParams {
  f3 origin, llc, hor, ver, u, v;
  float lens_radius;
};

// the same fields in binary64, as initCamera computes them (camera.nim:12-31): 19 doubles, in this order
struct CameraParams64 {
  double origin[3], llc[3], hor[3], ver[3], u[3], v[3];
  double lens_radius;
};

// Pixel jitter (raytracer.nim:41-42, u drawn before v) and thin-lens ray (camera.nim:34-39 with
// randomInUnitDisk, basetypes.nim:130-134).  i = column, j = row counted from the bottom.
RTOW_HD void camera_ray(const CameraParams& cam, int i, int j, int W, int H, Stream& g, f3& o, f3& d) {
  const float xi1 = draw(g), xi2 = draw(g);
  const float su = ((float)i + xi1) / (float)(W - 1);
  const float sv = ((float)j + xi2) / (float)(H - 1);
  float lx, ly;
  do { lx = draw_pm1(g); ly = draw_pm1(g); } while (fmaf(ly, ly, lx * lx) >= 1.0f);
  const float rdx = cam.lens_radius * lx, rdy = cam.lens_radius * ly;
  f3 off;
  off.x = fmaf(cam.v.x, rdy, cam.u.x * rdx);
  off.y = fmaf(cam.v.y, rdy, cam.u.y * rdx);
  off.z = fmaf(cam.v.z, rdy, cam.u.z * rdx);
  o.x = cam.origin.x + off.x; o.y = cam.origin.y + off.y; o.z = cam.origin.z + off.z;
  d.x = fmaf(sv, cam.ver.x, fmaf(su, cam.hor.x, cam.llc.x)) - cam.origin.x - off.x;
  d.y = fmaf(sv, cam.ver.y, fmaf(su, cam.hor.y, cam.llc.y)) - cam.origin.y - off.y;
  d.z = fmaf(sv, cam.ver.z, fmaf(su, cam.hor.z, cam.llc.z)) - cam.origin.z - off.z;
}

// Unit direction and the reference's t_min expressed as a distance.
RTOW_HD void segment_begin(f3 d, f3& u, float& tmin) {
  const float len = rt_sqrt(dot3(d, d));
  const float il = rt_rcp(len);
  u.x = d.x * il; u.y = d.y * il; u.z = d.z * il;
  tmin = kTMin * len;
}

// ------------------------------------------------------------------ candidate roots
// Sphere.hit (hitrecord.nim:55-79) with |dir| = 1: b = oc.u, disc = b^2 - (|oc|^2 - r^2) >= 0 on
// entry.  A sphere offers its near root if it is >= tmin, otherwise its far root if that is
// >= tmin (hitrecord.nim:66-71).  self: the ray starts on this sphere, so the near root is 0 by
// construction and only t = -2b is offered.
RTOW_HD bool regular_candidate(float b, float disc, float tmin, bool self, float& t) {
  // written with selects instead of branches: few lanes ever get here and the GPU runs both
  // sides of a branch anyway
  const float s = rt_sqrt(disc);
  const float t_near = -b - s, t_far = -b + s;
  t = t_near < tmin ? t_far : t_near;
  t = self ? -2.0f * b : t;
  return !(t < tmin);
}

// ------------------------------------------------------------------ the scan's filter
// Every sphere is first screened with  n^2 + (2o).c - k >= |o|^2 - margin,  n = c.u - o.u,
// k = |c|^2 - r^2 (binary64 on the host, rounded once), which is algebraically
// disc >= -margin.  margin = 2^-18 (|o|^2 + cmax^2) >= 2^-19 (|o| + cmax)^2 with cmax = max(|c| + |r|)
// over the table: at least 32 ulps of the largest intermediate, several times the rounding error
// the two evaluations can accumulate, and a few percent of r^2 for the scenes this renderer is
// used on.  Spheres that pass get the exact test; the screen is part of the arithmetic contract
// (the CPU mirror evaluates it with the same operations in the same order).
constexpr float kFilterMargin = 1.0f / 262144.0f;  // 2^-18

struct FilterRay {
  float nou;             // -(o.u)
  float ox2, oy2, oz2;   // 2*o
  float threshold;       // |o|^2 - margin
};

RTOW_HD FilterRay filter_ray(f3 o, f3 u, float cmax) {
  FilterRay f;
  f.nou = -dot3(o, u);
  f.ox2 = 2.0f * o.x; f.oy2 = 2.0f * o.y; f.oz2 = 2.0f * o.z;
  const float oo = dot3(o, o);
  f.threshold = oo - kFilterMargin * (oo + cmax * cmax);
  return f;
}

// filter value of one sphere (the kernels evaluate two at a time with packed instructions)
RTOW_HD float filter_value(float cx, float cy, float cz, float k, f3 u, const FilterRay& f) {
  float n = fmaf(cx, u.x, f.nou); n = fmaf(cy, u.y, n); n = fmaf(cz, u.z, n);
  float t = fmaf(cx, f.ox2, -k); t = fmaf(cy, f.oy2, t); t = fmaf(cz, f.oz2, t);
  return fmaf(n, n, t);
}

struct BigSphere {
  double cx, cy, cz, r;
  double rr;      // r*r
  int32_t index;  // position in the list
  int32_t pad_;
};

// Spheres with |r| >= kBigRadius: |oc|^2 - r^2 cancels catastrophically in binary32 (the ground
// sphere has r = 1000 and rays start within a few units of its surface), so oc, b = oc.u,
// c = |oc|^2 - r^2 and the discriminant are evaluated in binary64 from the binary32 ray.  The
// roots are then taken in binary32 in the cancellation-free form  q = -(b + sign(b) sqrt(disc)),
// {q, c/q}  (a = |u|^2 is 1 to rounding, as for every other sphere).
RTOW_HD bool big_candidate(f3 o, f3 u, const BigSphere& s, float tmin, bool self, float& t) {
  const double ux = u.x, uy = u.y, uz = u.z;
  const double ocx = (double)o.x - s.cx, ocy = (double)o.y - s.cy, ocz = (double)o.z - s.cz;
  const double hb = fma(ocz, uz, fma(ocy, uy, ocx * ux));
  const double c = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -s.rr)));
  const double disc = fma(hb, hb, -c);
  const float bf = (float)hb, cf = (float)c;
  const float sq = rt_sqrt((float)disc);
  const float q = bf < 0.0f ? sq - bf : -bf - sq;
  const float other = cf / q;
  const float t_near = fminf(q, other), t_far = fmaxf(q, other);
  t = t_near < tmin ? t_far : t_near;
  t = self ? -2.0f * bf : t;
  return disc >= 0.0 && !(t < tmin);
}

// closest-so-far update: smaller t wins, equal t goes to the later list index
// (hittables.nim:84-88 accepts root <= closestSoFar while scanning in list order)
RTOW_HD void offer(bool valid, float t, int j, float& best, int& id) {
  const bool better = valid && (t < best || (t == best && j > id));
  best = better ? t : best;
  id = better ? j : id;
}

// ------------------------------------------------------------------ shading
// r0 of Schlick's approximation (hitrecord.nim:103-104), squared, in binary32 as the CPU mirror
// evaluates it per hit; the device reads it from the sphere's record instead
RTOW_HD float schlick_r0(float ratio) {
  const float r0 = (1.0f - ratio) / (1.0f + ratio);
  return r0 * r0;
}

struct SphereShade {   // 48 bytes, fetched by hit index only
  float cx, cy, cz, inv_r;
  float ar, ag, ab, fuzz;   // albedo; for a dielectric (ar, ag) = schlick_r0(1/ir), schlick_r0(ir)
  int32_t kind; float ir, inv_ir, pad_;
};

// sky gradient (raytracer.nim:18-20) times throughput
RTOW_HD void shade_sky(f3 u, float tr, float tg, float tb, float& rr, float& rg, float& rb) {
  const float ts = 0.5f * (u.y + 1.0f);
  const float a = 1.0f - ts;
  rr = tr * fmaf(ts, 0.5f, a);
  rg = tg * fmaf(ts, 0.7f, a);
  rb = tb * fmaf(ts, 1.0f, a);
}

// Hit record (hitrecord.nim:73-77) + scatter (hitrecord.nim:82-132) at distance t along unit u.
// Updates origin, direction and throughput; returns false when the path is absorbed (Metal
// scattering below the surface, hitrecord.nim:99), in which case it contributes black.
// shade_at: the same at a hit point p that the caller has already computed.
// scatter (hitrecord.nim:82-132) on the material of record s, given the face-set unit normal n, the frontFace
// flag, the unit direction u of the incoming ray and - for Lambertian and Metal - the point of
// randomInUnitSphere (b, |b|^2 = l2) already drawn from g.  Multiplies the throughput by the attenuation.
// ALWAYS_OUT: also deliver direction and attenuation when the result is false (Metal scattering below the
// surface, hitrecord.nim:99), as the reference's var parameters do; the path tracer has no use for them.
template <bool ALWAYS_OUT>
RTOW_HD bool scatter_ball(const SphereShade& s, f3 n, bool front, f3 u, Stream& g, f3 b, float l2, f3& d_next,
                          float& tr, float& tg, float& tb) {
  f3 nd;
  if (s.kind == kDielectric) {  // hitrecord.nim:107-125
    const float ratio = front ? s.inv_ir : s.ir;
    const float dn = dot3(u, n);
    const float cos_t = fminf(-dn, 1.0f);
    const float sin_t = rt_sqrt(fmaf(-cos_t, cos_t, 1.0f));
    bool reflect_it = ratio * sin_t > 1.0f;
    if (!reflect_it) {  // the draw happens only when refraction is possible
      const float r0 = front ? s.ar : s.ag;   // schlick_r0(ratio), precomputed per sphere
      const float x = 1.0f - cos_t;
      const float x2 = x * x;
      const float x5 = x2 * x2 * x;
      const float refl = fmaf(1.0f - r0, x5, r0);  // Schlick, hitrecord.nim:101-105
      reflect_it = refl > draw(g);
    }
    if (reflect_it) {  // reflect, basetypes.nim:101-102
      const float k2 = -2.0f * dn;
      nd.x = fmaf(k2, n.x, u.x); nd.y = fmaf(k2, n.y, u.y); nd.z = fmaf(k2, n.z, u.z);
    } else {           // refract, basetypes.nim:104-108
      const f3 perp = {ratio * fmaf(cos_t, n.x, u.x), ratio * fmaf(cos_t, n.y, u.y),
                       ratio * fmaf(cos_t, n.z, u.z)};
      const float par = -rt_sqrt(fabsf(1.0f - dot3(perp, perp)));
      nd.x = fmaf(par, n.x, perp.x); nd.y = fmaf(par, n.y, perp.y); nd.z = fmaf(par, n.z, perp.z);
    }
  } else {
    if (s.kind == kLambertian) {  // hitrecord.nim:82-92
      const float ib = rt_rcp(rt_sqrt(l2));
      nd.x = fmaf(b.x, ib, n.x); nd.y = fmaf(b.y, ib, n.y); nd.z = fmaf(b.z, ib, n.z);
      if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;  // nearZero
    } else {                      // Metal, hitrecord.nim:94-99
      const float k2 = -2.0f * dot3(u, n);
      nd.x = fmaf(s.fuzz, b.x, fmaf(k2, n.x, u.x));
      nd.y = fmaf(s.fuzz, b.y, fmaf(k2, n.y, u.y));
      nd.z = fmaf(s.fuzz, b.z, fmaf(k2, n.z, u.z));
      if (!(dot3(nd, n) > 0.0f)) {
        if (ALWAYS_OUT) { tr *= s.ar; tg *= s.ag; tb *= s.ab; d_next = nd; }
        return false;
      }
    }
    tr *= s.ar; tg *= s.ag; tb *= s.ab;
  }
  d_next = nd;
  return true;
}

// shade_at_ball: shade_at with the point of randomInUnitSphere (b, |b|^2 = l2) already drawn from g by the
// caller; a dielectric ignores it (and must not have had it drawn).
RTOW_HD bool shade_at_ball(const SphereShade& s, f3 p, f3 u, Stream& g, f3 b, float l2, f3& d_next,
                           float& tr, float& tg, float& tb) {
  f3 n = {(p.x - s.cx) * s.inv_r, (p.y - s.cy) * s.inv_r, (p.z - s.cz) * s.inv_r};
  const bool front = dot3(u, n) < 0.0f;  // setFaceNormal, hitrecord.nim:51-53
  if (!front) { n.x = -n.x; n.y = -n.y; n.z = -n.z; }
  return scatter_ball<false>(s, n, front, u, g, b, l2, d_next, tr, tg, tb);
}

RTOW_HD bool shade_at(const SphereShade& s, f3 p, f3 u, Stream& g, f3& d_next, float& tr, float& tg, float& tb) {
  f3 b = {0.0f, 0.0f, 0.0f};
  float l2 = 0.0f;
  if (s.kind != kDielectric) b = unit_ball(g, l2);
  return shade_at_ball(s, p, u, g, b, l2, d_next, tr, tg, tb);
}

RTOW_HD bool shade_hit(const SphereShade& s, f3& o, f3 u, float t, Stream& g, f3& d_next,
                       float& tr, float& tg, float& tb) {
  const f3 p = {fmaf(t, u.x, o.x), fmaf(t, u.y, o.y), fmaf(t, u.z, o.z)};
  o = p;   // the scattered ray starts at the hit point whether or not it survives
  return shade_at(s, p, u, g, d_next, tr, tg, tb);
}

// one binary32 sample's channel radiance -> fixed point with 30 fractional bits; widen_fix() puts it in sum units
RTOW_HD uint32_t to_fix(float c) {
  c = fminf(fmaxf(c, 0.0f), kRadianceClamp);
  return (uint32_t)(c * kFixOne);
}
RTOW_HD uint64_t widen_fix(uint32_t v) { return (uint64_t)v << kSumShift; }

// a sample evaluated in binary64 (the sky colour of a primary miss) -> sum units (40 fractional bits, truncated)
RTOW_HD uint64_t to_fix64(double c) {
  c = c < 0.0 ? 0.0 : (c > (double)kRadianceClamp ? (double)kRadianceClamp : c);
  return (uint64_t)(c * kSumOneD);
}

// writeColor (raytracer.nim:22-28) on a fixed-point channel sum: mean, sqrt gamma,
// clamp(0, 0.999), 256x, truncate.  Runs in binary64.
RTOW_HD int quantise_fix(uint64_t sum, int spp) {
  const double mean = (double)sum / ((double)spp * kSumOneD);
  double v = sqrt(mean);
  v = v < 0.0 ? 0.0 : (v > 0.999 ? 0.999 : v);
  return (int)(256.0 * v);
}

}  // namespace rtow
