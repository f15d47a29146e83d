/* TEST INFRASTRUCTURE (see rtow_oracle.h).
 *
 * Faithful FP64 restatement of the reference's hot path, one C function per Nim proc, keeping
 * the reference's operation order (Nim `float` is 64-bit: basetypes.nim:8).  Compile with
 * -ffp-contract=off so no product-sum is fused behind the reference's back.
 */
#include "rtow_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double e[3]; } v3;

/* ---------------------------------------------------------------- basetypes.nim:17-108 */
static inline v3 mk(double x, double y, double z) { v3 r = {{x, y, z}}; return r; }
static inline v3 ld(const double* p) { return mk(p[0], p[1], p[2]); }
static inline void st(double* p, v3 a) { p[0] = a.e[0]; p[1] = a.e[1]; p[2] = a.e[2]; }
/* dot: accumulates from 0.0 in index order (basetypes.nim:25-27) */
static inline double dot(v3 v, v3 u) {
  double r = 0.0;
  for (int i = 0; i < 3; ++i) r += v.e[i] * u.e[i];
  return r;
}
/* cross (basetypes.nim:29-32) */
static inline v3 cross(v3 v, v3 u) {
  return mk(v.e[1] * u.e[2] - v.e[2] * u.e[1], v.e[2] * u.e[0] - v.e[0] * u.e[2],
            v.e[0] * u.e[1] - v.e[1] * u.e[0]);
}
static inline v3 neg(v3 v) { return mk(-v.e[0], -v.e[1], -v.e[2]); }
/* makeMath (basetypes.nim:68-81) */
static inline v3 add(v3 a, v3 b) { return mk(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline v3 sub(v3 a, v3 b) { return mk(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline v3 mulv(v3 a, v3 b) { return mk(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
static inline v3 smul(double s, v3 a) { return mk(s * a.e[0], s * a.e[1], s * a.e[2]); }
static inline v3 vmuls(v3 a, double s) { return mk(a.e[0] * s, a.e[1] * s, a.e[2] * s); }
static inline v3 vdivs(v3 a, double s) { return mk(a.e[0] / s, a.e[1] / s, a.e[2] / s); }
/* length_squared / length / normalize (basetypes.nim:87-94) */
static inline double length_squared(v3 v) { return v.e[0] * v.e[0] + v.e[1] * v.e[1] + v.e[2] * v.e[2]; }
static inline double length(v3 v) { return sqrt(length_squared(v)); }
static inline v3 unit_vector(v3 v) { return vdivs(v, length(v)); }
/* nearZero (basetypes.nim:96-99) */
static inline int near_zero(v3 v) {
  const double eps = 1e-8;
  return fabs(v.e[0]) < eps && fabs(v.e[1]) < eps && fabs(v.e[2]) < eps;
}
/* reflect (basetypes.nim:101-102): v - (2 * v.dot(n)) * n */
static inline v3 reflect(v3 v, v3 n) { return sub(v, smul(2 * dot(v, n), n)); }
/* refract (basetypes.nim:104-108) */
static inline v3 refract(v3 uv, v3 n, double etai_over_etat) {
  const double cos_theta = fmin(dot(neg(uv), n), 1.0);
  const v3 r_out_perp = smul(etai_over_etat, add(uv, smul(cos_theta, n)));
  const v3 r_out_parallel = smul(-sqrt(fabs(1.0 - length_squared(r_out_perp))), n);
  return add(r_out_perp, r_out_parallel);
}

/* ---------------------------------------------------------------- random sources */
typedef struct {
  int mode;
  orc_nimrand* nim;     /* ORC_RNG_NIM_SEQUENTIAL: the one global stream */
  uint32_t key[2];      /* ORC_RNG_COUNTER: stream of the current (pixel, sample) */
  uint32_t k;
} rng_t;

/* rand(max: float) */
static inline double rand_max(rng_t* g, double max) {
  if (g->mode == ORC_RNG_NIM_SEQUENTIAL) return orc_nimrand_rand(g->nim, max);
  return orc_draw_unit(g->key, g->k++) * max;
}
/* rand(a .. b) = rand(b - a) + a */
static inline double rand_range(rng_t* g, double a, double b) { return rand_max(g, b - a) + a; }

/* randomVec (basetypes.nim:110-112): x, y, z drawn in this order */
static inline v3 random_vec(rng_t* g, double lo, double hi) {
  const double x = rand_range(g, lo, hi);
  const double y = rand_range(g, lo, hi);
  const double z = rand_range(g, lo, hi);
  return mk(x, y, z);
}
/* randomInUnitSphere (basetypes.nim:114-118) */
static inline v3 random_in_unit_sphere(rng_t* g) {
  for (;;) {
    const v3 p = random_vec(g, -1, 1);
    if (length_squared(p) >= 1) continue;
    return p;
  }
}
/* randomUnitVector (basetypes.nim:120-121) */
static inline v3 random_unit_vector(rng_t* g) { return unit_vector(random_in_unit_sphere(g)); }
/* randomInUnitDisk (basetypes.nim:130-134) */
static inline v3 random_in_unit_disk(rng_t* g) {
  for (;;) {
    const double x = rand_range(g, -1.0, 1.0);
    const double y = rand_range(g, -1.0, 1.0);
    const v3 p = mk(x, y, 0);
    if (length_squared(p) >= 1.0) continue;
    return p;
  }
}

/* ---------------------------------------------------------------- hitrecord.nim */
typedef struct { v3 orig, dir; } ray_t;
typedef struct {
  v3 p, normal;
  double t;
  int front_face;
  const orc_sphere* mat; /* the Material copied at hitrecord.nim:77 (here: a pointer to its owner) */
  int index;
} hit_record;

static inline v3 ray_at(ray_t r, double t) { return add(r.orig, smul(t, r.dir)); } /* basetypes.nim:170 */

/* setFaceNormal (hitrecord.nim:51-53) */
static inline void set_face_normal(hit_record* rec, ray_t r, v3 outward_normal) {
  rec->front_face = dot(r.dir, outward_normal) < 0;
  rec->normal = rec->front_face ? outward_normal : neg(outward_normal);
}

/* Sphere.hit (hitrecord.nim:55-79) */
static int sphere_hit(const orc_sphere* s, ray_t r, double t_min, double t_max, hit_record* rec) {
  const v3 center = ld(s->center);
  const v3 oc = sub(r.orig, center);
  const double a = length_squared(r.dir);
  const double half_b = dot(oc, r.dir);
  const double c = length_squared(oc) - s->radius * s->radius;

  const double discriminant = half_b * half_b - a * c;
  if (discriminant < 0) return 0;
  const double sqrtd = sqrt(discriminant);

  double root = (-half_b - sqrtd) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || t_max < root) return 0;
  }
  rec->t = root;
  rec->p = ray_at(r, rec->t);
  const v3 outward_normal = vdivs(sub(rec->p, center), s->radius);
  set_face_normal(rec, r, outward_normal);
  rec->mat = s;
  return 1;
}

/* HittablesList.hit (hittables.nim:79-88): linear scan, ties go to the later element */
static int world_hit(const orc_sphere* w, int n, ray_t r, double t_min, double t_max, hit_record* rec) {
  hit_record tmp;
  int result = 0;
  double closest_so_far = t_max;
  for (int i = 0; i < n; ++i) {
    if (sphere_hit(&w[i], r, t_min, closest_so_far, &tmp)) {
      result = 1;
      closest_so_far = tmp.t;
      tmp.index = i;
      *rec = tmp;
    }
  }
  return result;
}

/* reflectance (hitrecord.nim:101-105) */
static inline double reflectance(double cosine, double ref_idx) {
  double r0 = (1 - ref_idx) / (1 + ref_idx);
  r0 = r0 * r0;
  return r0 + (1 - r0) * pow(1 - cosine, 5);
}

/* Negative controls for the tests that pin this file on the reference's own image: a deliberately WRONG
 * material model must fail those tests.  0 = faithful (the default, and the only value anything but
 * that one test uses); 1 = no Schlick reflection; 2 = Metal fuzz ignored; 3 = refraction ratio inverted. */
static atomic_int g_variant;
void orc_ref64_set_variant(int v) { atomic_store(&g_variant, v); }

/* scatter dispatch (hitrecord.nim:127-132) over the three material procs (:82-125) */
static int scatter(const orc_sphere* m, ray_t r_in, const hit_record* rec, rng_t* g, v3* attenuation,
                   ray_t* scattered) {
  switch (m->kind) {
    case ORC_LAMBERTIAN: { /* hitrecord.nim:82-92 */
      v3 scatter_direction = add(rec->normal, random_unit_vector(g));
      if (near_zero(scatter_direction)) scatter_direction = rec->normal;
      scattered->orig = rec->p;
      scattered->dir = scatter_direction;
      *attenuation = ld(m->albedo);
      return 1;
    }
    case ORC_METAL: { /* hitrecord.nim:94-99 */
      const v3 reflected = reflect(unit_vector(r_in.dir), rec->normal);
      scattered->orig = rec->p;
      scattered->dir = add(reflected, smul(atomic_load(&g_variant) == 2 ? 0.0 : m->fuzz, random_in_unit_sphere(g)));
      *attenuation = ld(m->albedo);
      return dot(scattered->dir, rec->normal) > 0;
    }
    default: { /* Dielectric, hitrecord.nim:107-125 */
      *attenuation = mk(1.0, 1.0, 1.0);
      const int variant = atomic_load(&g_variant);
      const double refraction_ratio = (rec->front_face != (variant == 3)) ? (1.0 / m->ir) : m->ir;
      const v3 unit_direction = unit_vector(r_in.dir);
      const double cos_theta = fmin(dot(neg(unit_direction), rec->normal), 1.0);
      const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
      const int cannot_refract = refraction_ratio * sin_theta > 1.0;
      v3 direction;
      /* Nim `or` short-circuits: the draw happens only when refraction is possible */
      if (cannot_refract || (variant != 1 ? reflectance(cos_theta, refraction_ratio) : 0.0) > rand_max(g, 1.0))
        direction = reflect(unit_direction, rec->normal);
      else
        direction = refract(unit_direction, rec->normal, refraction_ratio);
      scattered->orig = rec->p;
      scattered->dir = direction;
      return 1;
    }
  }
}

/* ---------------------------------------------------------------- raytracer.nim */
typedef struct {
  const orc_sphere* world;
  int n;
  uint64_t segments;
  int32_t* trail; /* optional: hit index of every segment of the current path (-1 = sky) */
  int trail_cap, trail_len;
} trace_ctx;
static inline void trail_push(trace_ctx* cx, int id) {
  if (cx->trail && cx->trail_len < cx->trail_cap) cx->trail[cx->trail_len] = id;
  if (cx->trail) cx->trail_len++;
}

/* rayColor (raytracer.nim:5-20), recursion kept */
static v3 ray_color(trace_ctx* cx, ray_t r, int depth, rng_t* g, int* first_id) {
  hit_record rec;
  if (depth <= 0) return mk(0, 0, 0);
  cx->segments++;
  if (world_hit(cx->world, cx->n, r, 0.001, INFINITY, &rec)) {
    if (first_id) *first_id = rec.index;
    trail_push(cx, rec.index);
    ray_t scattered;
    v3 attenuation;
    if (scatter(rec.mat, r, &rec, g, &attenuation, &scattered))
      return mulv(attenuation, ray_color(cx, scattered, depth - 1, g, NULL));
    return mk(0, 0, 0);
  }
  if (first_id) *first_id = -1;
  trail_push(cx, -1);
  const v3 unit_direction = unit_vector(r.dir);
  const double t = 0.5 * (unit_direction.e[1] + 1.0);
  return add(smul(1.0 - t, mk(1.0, 1.0, 1.0)), smul(t, mk(0.5, 0.7, 1.0)));
}

/* getRay (camera.nim:34-39) */
static ray_t get_ray(const orc_camera* c, double s, double t, rng_t* g) {
  const v3 rd = smul(c->lens_radius, random_in_unit_disk(g));
  const v3 offset = add(vmuls(ld(c->u), rd.e[0]), vmuls(ld(c->v), rd.e[1]));
  ray_t r;
  r.orig = add(ld(c->origin), offset);
  r.dir = sub(sub(add(add(ld(c->lower_left), smul(s, ld(c->horizontal))), smul(t, ld(c->vertical))),
                  ld(c->origin)),
              offset);
  return r;
}

/* writeColor (raytracer.nim:22-28): one channel */
int orc_ref64_quantise(double channel_sum, int spp) {
  const double scale = 1.0 / (double)spp;
  double r = sqrt(channel_sum * scale);
  r = r < 0 ? 0 : (r > 0.999 ? 0.999 : r); /* clamp(0, 0.999) */
  return (int)(256 * r);
}

typedef struct {
  const orc_sphere* spheres; int n; const orc_camera* cam;
  int W, H, spp, max_depth; uint64_t seed; int rng_mode; orc_nimrand* nim;
  const int32_t* rows; int n_rows; atomic_int* next_row;
  uint8_t* rgb; double* accum; int32_t* hit_id; uint64_t segments;
} job64;

/* the j/i/s loops of render (raytracer.nim:36-45); threads pull output rows from a shared counter */
static void* render_rows64(void* arg) {
  job64* jb = (job64*)arg;
  trace_ctx cx = {jb->spheres, jb->n, 0, NULL, 0, 0};
  rng_t g;
  g.mode = jb->rng_mode; g.nim = jb->nim; g.k = 0; g.key[0] = g.key[1] = 0;
  for (;;) {
    const int r = atomic_fetch_add(jb->next_row, 1);
    if (r >= jb->n_rows) break;
    const int y = jb->rows ? jb->rows[r] : r; /* 0 = top row = j = H-1 */
    const int j = jb->H - 1 - y;
    for (int i = 0; i < jb->W; ++i) {
      v3 pixel = mk(0, 0, 0);
      int first = -1;
      for (int s = 0; s < jb->spp; ++s) {
        if (g.mode == ORC_RNG_COUNTER) {
          orc_path_key(jb->seed, (uint32_t)(y * jb->W + i), (uint32_t)s, g.key);
          g.k = 0;
        }
        const double u = ((double)i + rand_max(&g, 1.0)) / (double)(jb->W - 1);
        const double v = ((double)j + rand_max(&g, 1.0)) / (double)(jb->H - 1);
        const ray_t ray = get_ray(jb->cam, u, v, &g);
        int id = -1;
        pixel = add(pixel, ray_color(&cx, ray, jb->max_depth, &g, s == 0 ? &id : NULL));
        if (s == 0) first = id;
      }
      const size_t o = (size_t)r * jb->W + i;
      if (jb->rgb)
        for (int c = 0; c < 3; ++c) jb->rgb[3 * o + c] = (uint8_t)orc_ref64_quantise(pixel.e[c], jb->spp);
      if (jb->accum) st(&jb->accum[3 * o], pixel);
      if (jb->hit_id) jb->hit_id[o] = first;
    }
  }
  jb->segments = cx.segments;
  return NULL;
}

int orc_ref64_render(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                     int spp, int max_depth, uint64_t seed, int rng_mode, orc_nimrand* nim_state,
                     const int32_t* rows, int n_rows, int n_threads, uint8_t* rgb, double* accum,
                     int32_t* hit_id, uint64_t* ray_segments) {
  if (W < 2 || H < 2 || spp < 1 || n_rows < 0 || n_spheres < 0) return -1;
  if (rng_mode == ORC_RNG_NIM_SEQUENTIAL) {
    if (!nim_state) return -1;
    n_threads = 1; /* one global stream: inherently serial */
  }
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_rows) n_threads = n_rows > 0 ? n_rows : 1;
  job64* jobs = (job64*)calloc((size_t)n_threads, sizeof(job64));
  pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
  atomic_int next_row = 0;
  uint64_t total = 0;
  for (int t = 0; t < n_threads; ++t) {
    job64 jb = {spheres, n_spheres, cam, W, H, spp, max_depth, seed, rng_mode, nim_state, rows,
                n_rows, &next_row, rgb, accum, hit_id, 0};
    jobs[t] = jb;
  }
  if (n_threads == 1) {
    render_rows64(&jobs[0]);
  } else {
    for (int t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, render_rows64, &jobs[t]);
    for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
  }
  for (int t = 0; t < n_threads; ++t) total += jobs[t].segments;
  if (ray_segments) *ray_segments = total;
  free(jobs);
  free(th);
  return 0;
}

/* ---------------------------------------------------------------- camera.nim:12-31 */
void orc_init_camera(const double look_from[3], const double look_at[3], const double vup[3],
                     double vfov_deg, double aspect, double aperture, double focus_dist,
                     orc_camera* cam) {
  const double theta = vfov_deg * (3.14159265358979323846 / 180.0); /* degToRad */
  const double h = tan(theta / 2.0);
  const double view_height = 2.0 * h;
  const double view_width = aspect * view_height;
  const v3 w = unit_vector(sub(ld(look_from), ld(look_at)));
  const v3 u = unit_vector(cross(ld(vup), w));
  const v3 v = cross(w, u);
  const v3 origin = ld(look_from);
  const v3 horizontal = smul(focus_dist * view_width, u);
  const v3 vertical = smul(focus_dist * view_height, v);
  const v3 llc = sub(sub(sub(origin, vdivs(horizontal, 2.0)), vdivs(vertical, 2.0)), smul(focus_dist, w));
  st(cam->origin, origin); st(cam->lower_left, llc); st(cam->horizontal, horizontal);
  st(cam->vertical, vertical); st(cam->u, u); st(cam->v, v); st(cam->w, w);
  cam->lens_radius = aperture / 2.0;
}

/* ---------------------------------------------------------------- scenes */
static orc_sphere mk_sphere(double x, double y, double z, double r, int kind, v3 albedo, double fuzz, double ir) {
  orc_sphere s;
  memset(&s, 0, sizeof s);
  s.center[0] = x; s.center[1] = y; s.center[2] = z; s.radius = r; s.kind = kind;
  st(s.albedo, albedo); s.fuzz = fuzz; s.ir = ir;
  return s;
}
#define PUSH(sp) do { if (n < cap) out[n] = (sp); ++n; } while (0)

/* sceneRedBlue (raytracer.nim:48-59) */
int orc_scene_red_blue(orc_sphere* out, int cap) {
  int n = 0;
  const double R = cos(3.14159265358979323846 / 4.0);
  PUSH(mk_sphere(-R, 0, -1, R, ORC_LAMBERTIAN, mk(0, 0, 1), 0, 0));
  PUSH(mk_sphere(R, 0, -1, R, ORC_LAMBERTIAN, mk(1, 0, 0), 0, 0));
  return n;
}

/* mixOfSpheres (raytracer.nim:61-73); note the negative-radius inner glass shell (:72) */
int orc_scene_mix_of_spheres(orc_sphere* out, int cap) {
  int n = 0;
  PUSH(mk_sphere(0, -100.5, -1, 100, ORC_LAMBERTIAN, mk(0.8, 0.8, 0.0), 0, 0));
  PUSH(mk_sphere(0, 0, -1, 0.5, ORC_LAMBERTIAN, mk(0.1, 0.2, 0.5), 0, 0));
  PUSH(mk_sphere(-1.0, 0, -1, 0.5, ORC_DIELECTRIC, mk(0, 0, 0), 0, 1.5));
  PUSH(mk_sphere(-1.0, 0, -1, -0.4, ORC_DIELECTRIC, mk(0, 0, 0), 0, 1.5));
  PUSH(mk_sphere(1.0, 0, -1, 0.5, ORC_METAL, mk(0.8, 0.6, 0.2), 1.0, 0));
  return n;
}

/* randomScene (raytracer.nim:75-111), grid half extent and material thresholds exposed */
int orc_scene_random(orc_nimrand* rng, int half_extent, double lambert_below, double metal_below,
                     orc_sphere* out, int cap, uint64_t* draws_out) {
  int n = 0;
  uint64_t draws = 0;
  rng_t g;
  g.mode = ORC_RNG_NIM_SEQUENTIAL; g.nim = rng; g.k = 0; g.key[0] = g.key[1] = 0;
#define RAND1(mx) (draws++, rand_max(&g, (mx)))
  PUSH(mk_sphere(0, -1000, 0, 1000, ORC_LAMBERTIAN, mk(0.5, 0.5, 0.5), 0, 0));
  for (int a = -half_extent; a < half_extent; ++a)
    for (int b = -half_extent; b < half_extent; ++b) {
      const double choose_mat = RAND1(1.0);
      const double cx = (double)a + 0.9 * RAND1(1.0);
      const double cz = (double)b + 0.9 * RAND1(1.0);
      const v3 center = mk(cx, 0.2, cz);
      if (length(sub(center, mk(4, 0.2, 0))) > 0.9) {
        if (choose_mat < lambert_below) {
          /* randomVec() = rand(0.0 .. 1.0) x3 = rand(1.0 - 0.0) + 0.0 */
          v3 a1, a2;
          for (int c = 0; c < 3; ++c) a1.e[c] = RAND1(1.0 - 0.0) + 0.0;
          for (int c = 0; c < 3; ++c) a2.e[c] = RAND1(1.0 - 0.0) + 0.0;
          PUSH(mk_sphere(cx, 0.2, cz, 0.2, ORC_LAMBERTIAN, mulv(a1, a2), 0, 0));
        } else if (choose_mat < metal_below) {
          v3 al;
          for (int c = 0; c < 3; ++c) al.e[c] = RAND1(1.0 - 0.5) + 0.5;
          const double fuzz = RAND1(0.5 - 0.0) + 0.0;
          PUSH(mk_sphere(cx, 0.2, cz, 0.2, ORC_METAL, al, fuzz, 0));
        } else {
          PUSH(mk_sphere(cx, 0.2, cz, 0.2, ORC_DIELECTRIC, mk(0, 0, 0), 0, 1.5));
        }
      }
    }
#undef RAND1
  PUSH(mk_sphere(0, 1, 0, 1.0, ORC_DIELECTRIC, mk(0, 0, 0), 0, 1.5));
  PUSH(mk_sphere(-4, 1, 0, 1.0, ORC_LAMBERTIAN, mk(0.4, 0.2, 0.1), 0, 0));
  PUSH(mk_sphere(4, 1, 0, 1.0, ORC_METAL, mk(0.7, 0.6, 0.5), 0.0, 0));
  if (draws_out) *draws_out = draws;
  return n;
}

/* ---------------------------------------------------------------- single-ray probes */
int orc_ref64_sphere_hit(const orc_sphere* s, const double orig[3], const double dir[3], double t_min,
                         double t_max, double* t, double p[3], double normal[3], int* front_face) {
  ray_t r = {ld(orig), ld(dir)};
  hit_record rec;
  if (!sphere_hit(s, r, t_min, t_max, &rec)) return 0;
  if (t) *t = rec.t;
  if (p) st(p, rec.p);
  if (normal) st(normal, rec.normal);
  if (front_face) *front_face = rec.front_face;
  return 1;
}

int orc_ref64_world_hit(const orc_sphere* spheres, int n, const double orig[3], const double dir[3],
                        double t_min, double t_max, double* t) {
  ray_t r = {ld(orig), ld(dir)};
  hit_record rec;
  if (!world_hit(spheres, n, r, t_min, t_max, &rec)) return -1;
  if (t) *t = rec.t;
  return rec.index;
}

int orc_ref64_scatter(const orc_sphere* s, const double in_orig[3], const double in_dir[3],
                      const double p[3], const double normal[3], int front_face, const uint32_t key[2],
                      uint32_t* k, double attenuation[3], double out_orig[3], double out_dir[3]) {
  ray_t r_in = {ld(in_orig), ld(in_dir)};
  hit_record rec;
  rec.p = ld(p); rec.normal = ld(normal); rec.t = 0; rec.front_face = front_face; rec.mat = s; rec.index = 0;
  rng_t g;
  g.mode = ORC_RNG_COUNTER; g.nim = NULL; g.key[0] = key[0]; g.key[1] = key[1]; g.k = *k;
  v3 att; ray_t sc;
  const int ok = scatter(s, r_in, &rec, &g, &att, &sc);
  *k = g.k;
  st(attenuation, att); st(out_orig, sc.orig); st(out_dir, sc.dir);
  return ok;
}

void orc_ref64_get_ray(const orc_camera* cam, double s, double t, const uint32_t key[2], uint32_t* k,
                       double orig[3], double dir[3]) {
  rng_t g;
  g.mode = ORC_RNG_COUNTER; g.nim = NULL; g.key[0] = key[0]; g.key[1] = key[1]; g.k = *k;
  const ray_t r = get_ray(cam, s, t, &g);
  *k = g.k;
  st(orig, r.orig); st(dir, r.dir);
}

/* hit-index trail and colour of one path (pixel column i, output row y, sample s), counter RNG */
int orc_ref64_trace_path(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                         int max_depth, uint64_t seed, int i, int y, int sample, int32_t* trail,
                         int trail_cap, double colour[3]) {
  trace_ctx cx = {spheres, n_spheres, 0, trail, trail_cap, 0};
  rng_t g;
  g.mode = ORC_RNG_COUNTER; g.nim = NULL; g.k = 0;
  orc_path_key(seed, (uint32_t)(y * W + i), (uint32_t)sample, g.key);
  const int j = H - 1 - y;
  const double u = ((double)i + rand_max(&g, 1.0)) / (double)(W - 1);
  const double v = ((double)j + rand_max(&g, 1.0)) / (double)(H - 1);
  const ray_t ray = get_ray(cam, u, v, &g);
  const v3 c = ray_color(&cx, ray, max_depth, &g, NULL);
  if (colour) st(colour, c);
  return cx.trail_len;
}

/* rayColor (raytracer.nim:5-20) of one caller-given ray, counter stream of path (pixel = index, sample 0) */
void orc_ref64_ray_color(const orc_sphere* spheres, int n, const double orig[3], const double dir[3], int depth,
                         uint64_t seed, uint32_t index, double colour[3]) {
  trace_ctx cx = {spheres, n, 0, NULL, 0, 0};
  rng_t g;
  g.mode = ORC_RNG_COUNTER; g.nim = NULL; g.k = 0;
  orc_path_key(seed, index, 0u, g.key);
  const ray_t r = {ld(orig), ld(dir)};
  st(colour, ray_color(&cx, r, depth, &g, NULL));
}
