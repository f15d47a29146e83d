/* TEST INFRASTRUCTURE (see rtow_oracle.h).
 *
 * FP32 mirror: an independent CPU restatement of the arithmetic the HIP kernels perform for
 * one path, written so that every rounding step is explicit.  GPU output must equal this
 * bit for bit (hit ids, fixed-point sums, 8-bit pixels, segment counts).
 *
 * The arithmetic contract (the kernels follow the same one):
 *   - every operation is a single IEEE-754 binary32 operation with round-to-nearest-even:
 *     +, -, *, /, sqrtf, and fmaf exactly where written; nothing else may be fused
 *     (compile with -ffp-contract=off);
 *   - dot3(a,b) = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x));
 *   - a ray direction is normalised once per segment, u = d * (1/sqrtf(dot3(d,d))), and the
 *     reference's t_min = 0.001 (raytracer.nim:11), which is in units of |d|, becomes
 *     0.001f*|d| in distance units;
 *   - rayColor's recursion (raytracer.nim:5-20) becomes a loop that carries the product of
 *     attenuations (throughput) outside-in;
 *   - spheres with |radius| >= 100 (the 1000-unit ground sphere, raytracer.nim:79) get oc, oc.u,
 *     |oc|^2 - r^2 and the discriminant in binary64 from the binary32 ray, because |oc|^2 - r^2
 *     cancels catastrophically in binary32; their roots are then taken in binary32 as
 *     q = -(b + sign(b) sqrt(disc)), {q, c/q}; all other spheres use binary32 only;
 *   - every binary32 sphere is first screened by a cheaper evaluation of its discriminant,
 *     n^2 + ((2o).c - k) >= |o|^2 - 2^-18 (|o|^2 + cmax^2) with n = c.u - o.u and k = |c|^2 - r^2,
 *     and only spheres that pass get the exact test below (the screen is there because the GPU
 *     evaluates it for two spheres per packed instruction; its margin is several times the
 *     rounding error of both evaluations, so in practice it only removes non-candidates);
 *   - a ray that starts on sphere j (it was scattered there) meets j again only at the
 *     analytic far root t = -2 (oc.u): the near root is 0 by construction and is dropped, which
 *     removes the binary32 self-intersections that t_min = 0.001|d| cannot catch when the
 *     scattered direction is short;
 *   - the PRIMARY segment of every path (camera ray, its closest hit, the sky colour of a miss) is
 *     evaluated with the reference's own binary64 operations in the reference's order
 *     (raytracer.nim:41-43, camera.nim:34-39, hitrecord.nim:55-71, hittables.nim:79-88,
 *     raytracer.nim:18-20), on the binary64 scene and camera, so that primary hit ids and the
 *     pixels of a one-sample depth-one image equal the reference arithmetic's.  The binary32 screen
 *     above still runs first, on the binary32 roundings of that ray and of the sphere table; only
 *     spheres that pass it (and the |radius| >= 100 ones) get the binary64 Sphere.hit.  The hit
 *     point p = orig + t*dir is rounded to binary32 and the path continues in binary32 from there
 *     (normal, scatter, throughput as for every later segment, with the binary32 unit direction
 *     of the rounded primary direction);
 *   - a uniform draw is (u32 >> 8) * 2^-24, so rand(-1.0..1.0) = 2*xi - 1 is exact;
 *   - per-sample radiance is accumulated per channel as a 64-bit integer with 40 fractional
 *     bits (order independent) - a binary32 sample is cut to 30 first, a binary64 sky sample keeps
 *     all 40 - and the writeColor step (raytracer.nim:22-28) runs in binary64 on that sum.
 */
#include "rtow_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } f3;

static inline float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

typedef struct {
  double cx64, cy64, cz64, r64; /* the scene description as given (primary segment) */
  float cx, cy, cz, r;  /* its binary32 roundings (every later segment, and the screen) */
  float r2;             /* r*r */
  float k;              /* |c|^2 - r^2, evaluated in binary64 and rounded once (filter operand) */
  float inv_r;          /* 1/r */
  int big;
  int kind;
  float ar, ag, ab;     /* albedo */
  float fuzz, ir, inv_ir;
} sph32;

typedef struct {
  f3 origin, llc, hor, ver, u, v;
  float lens_radius;
  const orc_camera* c64; /* the camera as given (primary rays are generated in binary64) */
} cam32;

#define BIG_RADIUS 100.0f
#define FILTER_MARGIN (1.0f / 262144.0f) /* 2^-18 */
#define FIX_ONE 1073741824.0f /* 2^30: scale of a binary32 sample */
#define SUM_SHIFT 10
#define SUM_ONE 1099511627776.0 /* 2^40: scale of the sums */
#define RADIANCE_CLAMP 3.5f

static void prep_scene(const orc_sphere* in, int n, sph32* out, float* cmax) {
  *cmax = 0.0f;
  for (int j = 0; j < n; ++j) {
    sph32 s;
    s.cx64 = in[j].center[0]; s.cy64 = in[j].center[1]; s.cz64 = in[j].center[2]; s.r64 = in[j].radius;
    s.cx = (float)in[j].center[0]; s.cy = (float)in[j].center[1]; s.cz = (float)in[j].center[2];
    s.r = (float)in[j].radius;
    s.big = fabsf(s.r) >= BIG_RADIUS;
    s.r2 = s.r * s.r;
    {
      const double c2 = (double)s.cx * s.cx + (double)s.cy * s.cy + (double)s.cz * s.cz;
      s.k = (float)(c2 - (double)s.r * s.r);
      if (!s.big) {
        const float reach = (float)sqrt(c2) + fabsf(s.r);
        if (reach > *cmax) *cmax = reach;
      }
    }
    s.inv_r = 1.0f / s.r;
    s.kind = in[j].kind;
    s.ar = (float)in[j].albedo[0]; s.ag = (float)in[j].albedo[1]; s.ab = (float)in[j].albedo[2];
    s.fuzz = (float)in[j].fuzz;
    s.ir = (float)in[j].ir;
    s.inv_ir = 1.0f / s.ir;
    out[j] = s;
  }
}

static f3 cvt3(const double* p) { f3 r = {(float)p[0], (float)p[1], (float)p[2]}; return r; }

static void prep_camera(const orc_camera* c, cam32* o) {
  o->origin = cvt3(c->origin); o->llc = cvt3(c->lower_left); o->hor = cvt3(c->horizontal);
  o->ver = cvt3(c->vertical); o->u = cvt3(c->u); o->v = cvt3(c->v);
  o->lens_radius = (float)c->lens_radius;
  o->c64 = c;
}

typedef struct { uint32_t key[2]; uint32_t k; } stream;
static inline float draw(stream* g) {
  return (float)(orc_draw_u32(g->key, g->k++) >> 8) * (1.0f / 16777216.0f);
}
static inline float draw_pm1(stream* g) { return fmaf(2.0f, draw(g), -1.0f); }

/* rejection sample of the open unit ball (basetypes.nim:114-118), returns |p|^2 too */
static inline f3 unit_ball(stream* g, float* l2_out) {
  for (;;) {
    f3 p;
    p.x = draw_pm1(g); p.y = draw_pm1(g); p.z = draw_pm1(g);
    const float l2 = dot3(p, p);
    if (l2 >= 1.0f) continue;
    *l2_out = l2;
    return p;
  }
}

/* Audit of the screen (tests only).  mode 0: normal.  mode 1: the screen is evaluated but never
 * rejects - every sphere gets the exact test - and every sphere whose exact discriminant is >= 0 is
 * tallied: how many of them the screen would have dropped (must stay 0), and how far above the
 * threshold the closest one sat, in units of the margin (1 = no rounding error at all, 0 = the
 * margin is used up). */
static atomic_int g_audit_mode;
static atomic_ullong g_audit_true_hits, g_audit_dropped;
static _Atomic double g_audit_min_slack;

void orc_mirror32_filter_audit(int mode) {
  atomic_store(&g_audit_mode, mode);
  atomic_store(&g_audit_true_hits, 0ull);
  atomic_store(&g_audit_dropped, 0ull);
  atomic_store(&g_audit_min_slack, 1e30);
}

void orc_mirror32_filter_audit_result(uint64_t* true_hits, uint64_t* dropped, double* min_slack_in_margins) {
  *true_hits = atomic_load(&g_audit_true_hits);
  *dropped = atomic_load(&g_audit_dropped);
  *min_slack_in_margins = atomic_load(&g_audit_min_slack);
}

static void audit_note(float filter_value, float threshold, float margin_abs) {
  atomic_fetch_add(&g_audit_true_hits, 1ull);
  if (!(filter_value >= threshold)) atomic_fetch_add(&g_audit_dropped, 1ull);
  const double slack = ((double)filter_value - (double)threshold) / (double)margin_abs;
  double cur = atomic_load(&g_audit_min_slack);
  while (slack < cur && !atomic_compare_exchange_weak(&g_audit_min_slack, &cur, slack)) { }
}

/* closest hit over the whole list: candidate of a sphere is its near root if >= tmin, else its
 * far root if >= tmin (hitrecord.nim:66-71); minimum wins, ties go to the later index
 * (hittables.nim:84-88) */
static int closest_hit(const sph32* w, int n, float cmax, f3 o, f3 u, float tmin, int prev, float* t_out) {
  float best = INFINITY;
  int id = -1;
  /* per-ray operands of the filter that screens every binary32 sphere:
   *   n^2 + (2o).c - k >= |o|^2 - margin,  n = c.u - o.u,  margin = 2^-18 (|o|^2 + cmax^2) */
  const float nou = -dot3(o, u);
  const float ox2 = 2.0f * o.x, oy2 = 2.0f * o.y, oz2 = 2.0f * o.z;
  const float oo = dot3(o, o);
  const float margin_abs = FILTER_MARGIN * (oo + cmax * cmax);
  const float threshold = oo - margin_abs;
  const int audit = atomic_load(&g_audit_mode);
  for (int j = 0; j < n; ++j) {
    float t;
    if (!w[j].big) {
      float fn = fmaf(w[j].cx, u.x, nou); fn = fmaf(w[j].cy, u.y, fn); fn = fmaf(w[j].cz, u.z, fn);
      float ft = fmaf(w[j].cx, ox2, -w[j].k); ft = fmaf(w[j].cy, oy2, ft); ft = fmaf(w[j].cz, oz2, ft);
      const float fv = fmaf(fn, fn, ft);
      if (!audit && !(fv >= threshold)) continue;
      const float ocx = o.x - w[j].cx, ocy = o.y - w[j].cy, ocz = o.z - w[j].cz;
      float b = ocx * u.x; b = fmaf(ocy, u.y, b); b = fmaf(ocz, u.z, b);
      float m = fmaf(-ocx, ocx, w[j].r2); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
      const float disc = fmaf(b, b, m);
      if (!(disc >= 0.0f)) continue;
      if (audit) audit_note(fv, threshold, margin_abs);
      if (j == prev) {
        t = -2.0f * b; /* origin lies on this sphere: roots are 0 and -2b */
        if (t < tmin) continue;
      } else {
        const float s = sqrtf(disc);
        t = -b - s;
        if (t < tmin) {
          t = -b + s;
          if (t < tmin) continue;
        }
      }
    } else {
      /* binary64 up to the discriminant, then binary32 roots in the cancellation-free form
       * q = -(b + sign(b) sqrt(disc)), {q, c/q} */
      const double ux = u.x, uy = u.y, uz = u.z;
      const double ocx = (double)o.x - (double)w[j].cx, ocy = (double)o.y - (double)w[j].cy,
                   ocz = (double)o.z - (double)w[j].cz;
      const double rr = (double)w[j].r * (double)w[j].r;
      const double hb = fma(ocz, uz, fma(ocy, uy, ocx * ux));
      const double c = fma(ocz, ocz, fma(ocy, ocy, fma(ocx, ocx, -rr)));
      const double disc = fma(hb, hb, -c);
      if (!(disc >= 0.0)) continue;
      const float bf = (float)hb, cf = (float)c;
      if (j == prev) {
        t = -2.0f * bf;
      } else {
        const float sq = sqrtf((float)disc);
        const float q = bf < 0.0f ? sq - bf : -bf - sq;
        const float other = cf / q;
        const float t_near = fminf(q, other), t_far = fmaxf(q, other);
        t = t_near < tmin ? t_far : t_near;
      }
      if (t < tmin) continue;
    }
    if (t <= best) { best = t; id = j; }
  }
  *t_out = best;
  return id;
}

/* Primary segment: HittablesList.hit (hittables.nim:79-88) over Sphere.hit (hitrecord.nim:55-71) in
 * binary64, in list order with the shrinking t_max, for the spheres the binary32 screen lets through
 * (o, u: binary32 roundings of the ray, u normalised in binary32) and for the |radius| >= 100 ones. */
static int primary_hit(const sph32* w, int n, float cmax, const double o64[3], const double d64[3], f3 o, f3 u,
                       double t_min, double t_max, double* t_out) {
  double closest = t_max;
  int id = -1;
  const float nou = -dot3(o, u);
  const float ox2 = 2.0f * o.x, oy2 = 2.0f * o.y, oz2 = 2.0f * o.z;
  const float oo = dot3(o, o);
  const float margin_abs = FILTER_MARGIN * (oo + cmax * cmax);
  const float threshold = oo - margin_abs;
  const int audit = atomic_load(&g_audit_mode);
  const double a = d64[0] * d64[0] + d64[1] * d64[1] + d64[2] * d64[2]; /* length_squared, basetypes.nim:88 */
  for (int j = 0; j < n; ++j) {
    float fv = 0.0f;
    if (!w[j].big) {
      float fn = fmaf(w[j].cx, u.x, nou); fn = fmaf(w[j].cy, u.y, fn); fn = fmaf(w[j].cz, u.z, fn);
      float ft = fmaf(w[j].cx, ox2, -w[j].k); ft = fmaf(w[j].cy, oy2, ft); ft = fmaf(w[j].cz, oz2, ft);
      fv = fmaf(fn, fn, ft);
      if (!audit && !(fv >= threshold)) continue;
    }
    const double ocx = o64[0] - w[j].cx64, ocy = o64[1] - w[j].cy64, ocz = o64[2] - w[j].cz64;
    double half_b = 0.0; /* dot accumulates from 0.0 in index order, basetypes.nim:25-27 */
    half_b += ocx * d64[0]; half_b += ocy * d64[1]; half_b += ocz * d64[2];
    const double c = (ocx * ocx + ocy * ocy + ocz * ocz) - w[j].r64 * w[j].r64;
    const double disc = half_b * half_b - a * c;
    if (disc < 0) continue;
    if (audit && !w[j].big) audit_note(fv, threshold, margin_abs);
    const double sqrtd = sqrt(disc);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || closest < root) {
      root = (-half_b + sqrtd) / a;
      if (root < t_min || closest < root) continue;
    }
    closest = root;
    id = j;
  }
  *t_out = closest;
  return id;
}

/* one channel of a binary64 sample -> sum units: fixed point, 40 fractional bits, truncated */
static inline uint64_t to_fix64(double c) {
  c = c < 0.0 ? 0.0 : (c > (double)RADIANCE_CLAMP ? (double)RADIANCE_CLAMP : c);
  return (uint64_t)(c * SUM_ONE);
}

typedef struct {
  const sph32* w; int n; float cmax; const cam32* cam;
  int W, H, spp, max_depth; uint64_t seed;
  const int32_t* rows; int n_rows; atomic_int* next_row;
  uint8_t* rgb; uint64_t* accum; int32_t* hit_id; uint64_t segments;
} job32;

/* one channel of a binary32 sample: cut to 30 fractional bits (it then fits 32 bits), widened to sum units */
static inline uint64_t to_fix(float c) {
  c = fminf(fmaxf(c, 0.0f), RADIANCE_CLAMP);
  return (uint64_t)(uint32_t)(c * FIX_ONE) << SUM_SHIFT;
}

/* scatter (hitrecord.nim:82-132) in binary32: n is the face-set unit normal, front the frontFace flag, u the
 * unit direction of the incoming ray.  Sets the next direction and multiplies the throughput by the
 * attenuation - also when it returns 0 (Metal scattering below the surface, hitrecord.nim:99). */
static int scatter32(const sph32* s, f3 n, int front, f3 u, stream* g, f3* nd_out, float* tr, float* tg, float* tb) {
  f3 nd;
  int ok = 1;
  if (s->kind == ORC_DIELECTRIC) { /* hitrecord.nim:107-125 */
    const float ratio = front ? s->inv_ir : s->ir;
    const float dn = dot3(u, n);
    const float cos_t = fminf(-dn, 1.0f);
    const float sin_t = sqrtf(fmaf(-cos_t, cos_t, 1.0f));
    int reflect_it = ratio * sin_t > 1.0f;
    if (!reflect_it) {
      float r0 = (1.0f - ratio) / (1.0f + ratio);
      r0 = r0 * r0;
      const float x = 1.0f - cos_t;
      const float x2 = x * x;
      const float x5 = x2 * x2 * x;
      const float refl = fmaf(1.0f - r0, x5, r0);
      reflect_it = refl > draw(g);
    }
    if (reflect_it) {
      const float k2 = -2.0f * dn;
      nd.x = fmaf(k2, n.x, u.x); nd.y = fmaf(k2, n.y, u.y); nd.z = fmaf(k2, n.z, u.z);
    } else {
      f3 perp = {ratio * fmaf(cos_t, n.x, u.x), ratio * fmaf(cos_t, n.y, u.y), ratio * fmaf(cos_t, n.z, u.z)};
      const float par = -sqrtf(fabsf(1.0f - dot3(perp, perp)));
      nd.x = fmaf(par, n.x, perp.x); nd.y = fmaf(par, n.y, perp.y); nd.z = fmaf(par, n.z, perp.z);
    }
    /* attenuation is white */
  } else {
    float l2;
    const f3 b = unit_ball(g, &l2);
    if (s->kind == ORC_LAMBERTIAN) { /* hitrecord.nim:82-92 */
      const float ib = 1.0f / sqrtf(l2);
      nd.x = fmaf(b.x, ib, n.x); nd.y = fmaf(b.y, ib, n.y); nd.z = fmaf(b.z, ib, n.z);
      if (fabsf(nd.x) < 1e-8f && fabsf(nd.y) < 1e-8f && fabsf(nd.z) < 1e-8f) nd = n;
    } else { /* Metal, hitrecord.nim:94-99 */
      const float k2 = -2.0f * dot3(u, n);
      nd.x = fmaf(s->fuzz, b.x, fmaf(k2, n.x, u.x));
      nd.y = fmaf(s->fuzz, b.y, fmaf(k2, n.y, u.y));
      nd.z = fmaf(s->fuzz, b.z, fmaf(k2, n.z, u.z));
      ok = dot3(nd, n) > 0.0f;
    }
    *tr *= s->ar; *tg *= s->ag; *tb *= s->ab;
  }
  *nd_out = nd;
  return ok;
}

/* rayColor (raytracer.nim:5-20) for one ray given in binary64: the first world.hit in binary64 over
 * [t_min, t_max], every later segment in binary32.  colour: binary64 sky colour for a first-segment miss,
 * otherwise the binary32 radiance of the path (widened); returns 1 when the colour is such a binary64 sky
 * sample.  Counts world.hit queries. */
static int trace_ray(const job32* jb, const double o64[3], const double d64[3], double t_min, double t_max, stream* gp,
                     int max_depth, double colour[3], int* first_id, double* first_t, uint64_t* segments, int32_t* trail,
                     int trail_cap, int* trail_len) {
  stream g = *gp;
  f3 o = {(float)o64[0], (float)o64[1], (float)o64[2]};
  f3 d = {(float)d64[0], (float)d64[1], (float)d64[2]};

  float tr = 1.0f, tg = 1.0f, tb = 1.0f; /* throughput */
  float rr = 0.0f, rg = 0.0f, rb = 0.0f; /* radiance of the path */
  *first_id = -1;
  if (first_t) *first_t = INFINITY;
  int prev = -1; /* sphere the ray starts on */
  colour[0] = colour[1] = colour[2] = 0.0;
  for (int depth = max_depth; depth > 0; --depth) { /* depth <= 0 -> black (raytracer.nim:8-9) */
    ++*segments;
    const float len = sqrtf(dot3(d, d));
    const float il = 1.0f / len;
    const f3 u = {d.x * il, d.y * il, d.z * il};
    const float tmin = 0.001f * len;
    float t = 0.0f;
    int id;
    f3 p;
    if (depth == max_depth) {
      double t64;
      id = primary_hit(jb->w, jb->n, jb->cmax, o64, d64, o, u, t_min, t_max, &t64);
      *first_id = id;
      if (first_t && id >= 0) *first_t = t64;
      if (trail) { if (*trail_len < trail_cap) trail[*trail_len] = id; ++*trail_len; }
      if (id < 0) { /* sky, raytracer.nim:18-20, in binary64; the throughput is 1 */
        const double a = d64[0] * d64[0] + d64[1] * d64[1] + d64[2] * d64[2];
        const double uy = d64[1] / sqrt(a);
        const double ts = 0.5 * (uy + 1.0);
        colour[0] = (1.0 - ts) * 1.0 + ts * 0.5;
        colour[1] = (1.0 - ts) * 1.0 + ts * 0.7;
        colour[2] = (1.0 - ts) * 1.0 + ts * 1.0;
        *gp = g;
        return 1;
      }
      p.x = (float)(o64[0] + t64 * d64[0]); p.y = (float)(o64[1] + t64 * d64[1]); p.z = (float)(o64[2] + t64 * d64[2]);
    } else {
      id = closest_hit(jb->w, jb->n, jb->cmax, o, u, tmin, prev, &t);
      if (trail) { if (*trail_len < trail_cap) trail[*trail_len] = id; ++*trail_len; }
      if (id < 0) { /* sky, raytracer.nim:18-20 */
        const float ts = 0.5f * (u.y + 1.0f);
        const float a = 1.0f - ts;
        rr = tr * fmaf(ts, 0.5f, a);
        rg = tg * fmaf(ts, 0.7f, a);
        rb = tb * fmaf(ts, 1.0f, a);
        break;
      }
      p.x = fmaf(t, u.x, o.x); p.y = fmaf(t, u.y, o.y); p.z = fmaf(t, u.z, o.z);
    }
    const sph32* s = &jb->w[id];
    f3 n = {(p.x - s->cx) * s->inv_r, (p.y - s->cy) * s->inv_r, (p.z - s->cz) * s->inv_r};
    const int front = dot3(u, n) < 0.0f;
    if (!front) { n.x = -n.x; n.y = -n.y; n.z = -n.z; }
    f3 nd; /* next direction */
    if (!scatter32(s, n, front, u, &g, &nd, &tr, &tg, &tb)) break; /* absorbed: black */
    o = p;
    d = nd;
    prev = id;
  }
  colour[0] = rr; colour[1] = rg; colour[2] = rb;
  *gp = g;
  return 0;
}

/* one path of the render: camera ray in binary64 (raytracer.nim:41-42 u before v, camera.nim:34-39 with
 * basetypes.nim:130-134; every draw is a 24-bit value, exactly representable in both widths), then
 * rayColor; returns radiance in fixed point */
static void trace_path(const job32* jb, int i, int j, uint32_t pixel, uint32_t sample, uint64_t fix[3],
                       int* first_id, uint64_t* segments, int32_t* trail, int trail_cap, int* trail_len) {
  const orc_camera* c64 = jb->cam->c64;
  stream g;
  orc_path_key(jb->seed, pixel, sample, g.key);
  g.k = 0;
  const double xi1 = (double)draw(&g), xi2 = (double)draw(&g);
  const double su = ((double)i + xi1) / (double)(jb->W - 1);
  const double sv = ((double)j + xi2) / (double)(jb->H - 1);
  double lx, ly;
  do { lx = (double)draw(&g) * 2.0 + -1.0; ly = (double)draw(&g) * 2.0 + -1.0; } while (lx * lx + ly * ly >= 1.0);
  const double rdx = c64->lens_radius * lx, rdy = c64->lens_radius * ly;
  double o64[3], d64[3];
  for (int k = 0; k < 3; ++k) {
    const double off = c64->u[k] * rdx + c64->v[k] * rdy;
    o64[k] = c64->origin[k] + off;
    d64[k] = c64->lower_left[k] + su * c64->horizontal[k] + sv * c64->vertical[k] - c64->origin[k] - off;
  }
  double colour[3];
  const int sky64 = trace_ray(jb, o64, d64, 0.001, INFINITY, &g, jb->max_depth, colour, first_id, NULL, segments, trail,
                              trail_cap, trail_len);
  for (int k = 0; k < 3; ++k) fix[k] = sky64 ? to_fix64(colour[k]) : to_fix((float)colour[k]);
}

static int quantise_fix(uint64_t sum, int spp) {
  const double mean = (double)sum / ((double)spp * SUM_ONE);
  double v = sqrt(mean);
  v = v < 0.0 ? 0.0 : (v > 0.999 ? 0.999 : v);
  return (int)(256.0 * v);
}

static void* render_rows32(void* arg) {
  job32* jb = (job32*)arg;
  uint64_t segments = 0;
  for (;;) {
    const int r = atomic_fetch_add(jb->next_row, 1);
    if (r >= jb->n_rows) break;
    const int y = jb->rows ? jb->rows[r] : r;
    const int j = jb->H - 1 - y;
    for (int i = 0; i < jb->W; ++i) {
      uint64_t sum[3] = {0, 0, 0};
      int first = -1;
      for (int s = 0; s < jb->spp; ++s) {
        uint64_t fix[3];
        int id;
        trace_path(jb, i, j, (uint32_t)(y * jb->W + i), (uint32_t)s, fix, &id, &segments, NULL, 0, NULL);
        if (s == 0) first = id;
        sum[0] += fix[0]; sum[1] += fix[1]; sum[2] += fix[2];
      }
      const size_t o = (size_t)r * jb->W + i;
      if (jb->rgb)
        for (int c = 0; c < 3; ++c) jb->rgb[3 * o + c] = (uint8_t)quantise_fix(sum[c], jb->spp);
      if (jb->accum) { jb->accum[3 * o] = sum[0]; jb->accum[3 * o + 1] = sum[1]; jb->accum[3 * o + 2] = sum[2]; }
      if (jb->hit_id) jb->hit_id[o] = first;
    }
  }
  jb->segments = segments;
  return NULL;
}

int orc_mirror32_render(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                        int spp, int max_depth, uint64_t seed, const int32_t* rows, int n_rows,
                        int n_threads, uint8_t* rgb, uint64_t* accum_fix, int32_t* hit_id,
                        uint64_t* ray_segments) {
  if (W < 2 || H < 2 || spp < 1 || n_rows < 0 || n_spheres < 0) return -1;
  sph32* w = (sph32*)malloc(sizeof(sph32) * (size_t)(n_spheres > 0 ? n_spheres : 1));
  float cmax;
  prep_scene(spheres, n_spheres, w, &cmax);
  cam32 c;
  prep_camera(cam, &c);
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_rows) n_threads = n_rows > 0 ? n_rows : 1;
  job32* jobs = (job32*)calloc((size_t)n_threads, sizeof(job32));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  atomic_int next_row = 0;
  for (int t = 0; t < n_threads; ++t) {
    job32 jb = {w, n_spheres, cmax, &c, W, H, spp, max_depth, seed, rows, n_rows, &next_row,
                rgb, accum_fix, hit_id, 0};
    jobs[t] = jb;
  }
  if (n_threads == 1) {
    render_rows32(&jobs[0]);
  } else {
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, render_rows32, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  uint64_t total = 0;
  for (int t = 0; t < n_threads; ++t) total += jobs[t].segments;
  if (ray_segments) *ray_segments = total;
  free(jobs); free(th); free(w);
  return 0;
}

/* hit-index trail and fixed-point colour of one path (pixel column i, output row y, sample s) */
int orc_mirror32_trace_path(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                            int max_depth, uint64_t seed, int i, int y, int sample, int32_t* trail,
                            int trail_cap, uint64_t fix[3]) {
  sph32* w = (sph32*)malloc(sizeof(sph32) * (size_t)(n_spheres > 0 ? n_spheres : 1));
  float cmax;
  prep_scene(spheres, n_spheres, w, &cmax);
  cam32 c;
  prep_camera(cam, &c);
  job32 jb = {w, n_spheres, cmax, &c, W, H, 1, max_depth, seed, NULL, 0, NULL, NULL, NULL, NULL, 0};
  int first, len = 0;
  uint64_t seg = 0, f[3];
  trace_path(&jb, i, H - 1 - y, (uint32_t)(y * W + i), (uint32_t)sample, f, &first, &seg, trail, trail_cap, &len);
  if (fix) { fix[0] = f[0]; fix[1] = f[1]; fix[2] = f[2]; }
  free(w);
  return len;
}

/* rayColor / hit of one caller-given ray with the renderer's arithmetic (what rtow_trace_rays computes):
 * returns the list index hit first in [t_min, t_max] (-1: none), its t, and the colour; the random stream
 * is that of path (pixel = index, sample 0), from its first draw */
int orc_mirror32_trace_ray(const orc_sphere* spheres, int n_spheres, const double orig[3], const double dir[3],
                           double t_min, double t_max, int max_depth, uint64_t seed, uint32_t index, double colour[3],
                           double* t) {
  sph32* w = (sph32*)malloc(sizeof(sph32) * (size_t)(n_spheres > 0 ? n_spheres : 1));
  float cmax;
  prep_scene(spheres, n_spheres, w, &cmax);
  job32 jb = {w, n_spheres, cmax, NULL, 2, 2, 1, max_depth, seed, NULL, 0, NULL, NULL, NULL, NULL, 0};
  stream g;
  orc_path_key(seed, index, 0u, g.key);
  g.k = 0;
  int first = -1, len = 0;
  uint64_t seg = 0;
  double c[3] = {0, 0, 0}, t64 = INFINITY;
  if (max_depth > 0) {
    (void)trace_ray(&jb, orig, dir, t_min, t_max, &g, max_depth, c, &first, &t64, &seg, NULL, 0, &len);
  } else { /* the hit query alone */
    f3 o = {(float)orig[0], (float)orig[1], (float)orig[2]};
    f3 d = {(float)dir[0], (float)dir[1], (float)dir[2]};
    const float il = 1.0f / sqrtf(dot3(d, d));
    const f3 u = {d.x * il, d.y * il, d.z * il};
    first = primary_hit(w, n_spheres, cmax, orig, dir, o, u, t_min, t_max, &t64);
    if (first < 0) t64 = INFINITY;
  }
  if (colour) { colour[0] = c[0]; colour[1] = c[1]; colour[2] = c[2]; }
  if (t) *t = t64;
  free(w);
  return first;
}

/* scatter for one (material, incoming ray, hit record) triple in the device's arithmetic: inputs rounded to
 * binary32, the direction normalised first; draws from the stream of path (pixel index, sample 0) */
int orc_mirror32_scatter(const orc_sphere* material, const double in_dir[3], const double normal[3], int front_face,
                         uint64_t seed, uint32_t index, double attenuation[3], double out_dir[3]) {
  sph32 w;
  float cmax;
  prep_scene(material, 1, &w, &cmax);
  const f3 d = {(float)in_dir[0], (float)in_dir[1], (float)in_dir[2]};
  const f3 n = {(float)normal[0], (float)normal[1], (float)normal[2]};
  const float il = 1.0f / sqrtf(dot3(d, d));
  const f3 u = {d.x * il, d.y * il, d.z * il};
  stream g;
  orc_path_key(seed, index, 0u, g.key);
  g.k = 0;
  float tr = 1.0f, tg = 1.0f, tb = 1.0f;
  f3 nd;
  const int ok = scatter32(&w, n, front_face != 0, u, &g, &nd, &tr, &tg, &tb);
  attenuation[0] = tr; attenuation[1] = tg; attenuation[2] = tb;
  out_dir[0] = nd.x; out_dir[1] = nd.y; out_dir[2] = nd.z;
  return ok;
}
