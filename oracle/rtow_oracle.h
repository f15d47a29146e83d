/* TEST INFRASTRUCTURE - not part of the shipped renderer.
 *
 * CPU oracle for the per-pixel hot path of Vindaar/rayTracingInOneWeekend
 * (render -> getRay -> rayColor -> HittablesList.hit -> Sphere.hit -> scatter -> writeColor).
 * Two independent restatements live behind this header:
 *
 *   ref64    faithful FP64 restatement of the Nim reference, function by function
 *            (recursive rayColor, unnormalised directions, a = |dir|^2, libm pow ...).
 *   mirror32 FP32 restatement of the arithmetic the GPU kernels perform (iterative
 *            throughput loop, unit directions, explicit fmaf order, fixed-point sums),
 *            used for bit-exact GPU-vs-CPU checks.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the renderer itself never does.
 *
 * Parity status: the reference ships no tests.  What pins this oracle is (i) the sky rows
 * and sphere layout of the reference's own output image media/rayTracing.png and (ii) the
 * golden values of the Nim std/random restatement recorded in tests/golden/.  The
 * sequential per-pixel noise of the reference (global RNG stream) is "parity unpinned".
 */
#ifndef RTOW_ORACLE_H
#define RTOW_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material kinds: hitrecord.nim:16-17 */
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_DIELECTRIC = 2 };

/* One sphere of a HittablesList (hitrecord.nim:11-14 + the Material it embeds, :18-32). */
typedef struct {
  double center[3];
  double radius;
  int32_t kind;
  int32_t pad_;
  double albedo[3]; /* Lambertian / Metal */
  double fuzz;      /* Metal */
  double ir;        /* Dielectric */
} orc_sphere;

/* camera.nim:4-10 */
typedef struct {
  double origin[3];
  double lower_left[3];
  double horizontal[3];
  double vertical[3];
  double u[3], v[3], w[3];
  double lens_radius;
} orc_camera;

/* ---- Nim std/random (xoroshiro128+) restatement; used by the scene builders ---- */
typedef struct { uint64_t a0, a1; } orc_nimrand;
void orc_nimrand_default(orc_nimrand* r);                  /* state of the un-seeded global generator */
void orc_nimrand_init(orc_nimrand* r, int64_t seed, int with_jump); /* randomize(seed) / initRand */
uint64_t orc_nimrand_next(orc_nimrand* r);
double orc_nimrand_rand(orc_nimrand* r, double max);       /* rand(max: float) */

/* ---- counter-based per-(pixel,sample) generator shared with the GPU ---- */
void orc_path_key(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t key[2]);
uint32_t orc_draw_u32(const uint32_t key[2], uint32_t k);   /* k-th raw draw of the path */
/* uniform in [0,1) with 24 random bits: exactly representable in FP32 and FP64 */
double orc_draw_unit(const uint32_t key[2], uint32_t k);

/* ---- scenes (raytracer.nim:48-111).  Return the sphere count, fill at most cap. ---- */
int orc_scene_red_blue(orc_sphere* out, int cap);
int orc_scene_mix_of_spheres(orc_sphere* out, int cap);
/* randomScene generalised: half_extent 11 and thresholds 0.8/0.95 give the reference scene.
 * rng is advanced exactly as the Nim code would; draws_out counts the calls to next(). */
int orc_scene_random(orc_nimrand* rng, int half_extent, double lambert_below, double metal_below,
                     orc_sphere* out, int cap, uint64_t* draws_out);

/* ---- camera.nim:12-31 ---- */
void orc_init_camera(const double look_from[3], const double look_at[3], const double vup[3],
                     double vfov_deg, double aspect, double aperture, double focus_dist,
                     orc_camera* cam);

/* rng_mode */
enum { ORC_RNG_COUNTER = 0, ORC_RNG_NIM_SEQUENTIAL = 1 };

/* Render rows[0..n_rows) (row 0 = top image row, the first row the reference writes,
 * raytracer.nim:36) of a W x H frame.  Outputs are n_rows x W, in rows[] order; any may be NULL.
 *   rgb      3 bytes per pixel, writeColor quantisation (raytracer.nim:22-28)
 *   accum    3 doubles per pixel: the un-scaled colour sum over samples
 *   hit_id   list index of the sphere hit by the primary ray of sample 0, -1 for sky
 *   ray_segments  number of world.hit queries issued (raytracer.nim:11)
 * ORC_RNG_NIM_SEQUENTIAL reproduces the reference's single global stream (rows must then be
 * 0..H-1 in order for the result to mean anything); nim_state is its in/out state.
 * n_threads > 1 splits rows over pthreads (counter mode only). */
int orc_ref64_render(const orc_sphere* spheres, int n_spheres, const orc_camera* cam,
                     int W, int H, int spp, int max_depth, uint64_t seed, int rng_mode,
                     orc_nimrand* nim_state, const int32_t* rows, int n_rows, int n_threads,
                     uint8_t* rgb, double* accum, int32_t* hit_id, uint64_t* ray_segments);

/* Test-only switch to a deliberately wrong material model (negative controls): 0 = faithful (default),
 * 1 = no Schlick reflection, 2 = Metal fuzz ignored, 3 = refraction ratio inverted.  Process-wide. */
void orc_ref64_set_variant(int v);

/* FP32 mirror of the GPU arithmetic.  accum_fix: 3 x uint64 per pixel, fixed-point sums with
 * 40 fractional bits (what the kernels accumulate). */
int orc_mirror32_render(const orc_sphere* spheres, int n_spheres, const orc_camera* cam,
                        int W, int H, int spp, int max_depth, uint64_t seed,
                        const int32_t* rows, int n_rows, int n_threads,
                        uint8_t* rgb, uint64_t* accum_fix, int32_t* hit_id, uint64_t* ray_segments);

/* Audit of the mirror's sphere screen (n^2 + (2o).c - k >= |o|^2 - margin), for tests.  mode 1:
 * the screen stops rejecting and every sphere with an exact discriminant >= 0 is tallied until
 * mode 0 is set again; _result reports how many there were, how many the screen would have
 * dropped (the renderer is only correct while that is 0) and the smallest (value - threshold) /
 * margin among them (1 = no rounding error, 0 = margin used up).  Process-wide state. */
void orc_mirror32_filter_audit(int mode);
void orc_mirror32_filter_audit_result(uint64_t* true_hits, uint64_t* dropped, double* min_slack_in_margins);

/* One path in isolation: writes the list index hit by every segment (-1 = sky) into trail[]
 * (at most trail_cap entries) and returns the number of segments. */
int orc_ref64_trace_path(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                         int max_depth, uint64_t seed, int i, int y, int sample, int32_t* trail,
                         int trail_cap, double colour[3]);
int orc_mirror32_trace_path(const orc_sphere* spheres, int n_spheres, const orc_camera* cam, int W, int H,
                            int max_depth, uint64_t seed, int i, int y, int sample, int32_t* trail,
                            int trail_cap, uint64_t fix[3]);

/* single-ray probes used by unit tests ------------------------------------------------ */
/* Sphere.hit (hitrecord.nim:55-79): returns 1 on hit and fills t, p, normal, front_face */
int orc_ref64_sphere_hit(const orc_sphere* s, const double orig[3], const double dir[3],
                         double t_min, double t_max, double* t, double p[3], double normal[3],
                         int* front_face);
/* HittablesList.hit (hittables.nim:79-88): returns hit index or -1 */
int orc_ref64_world_hit(const orc_sphere* spheres, int n, const double orig[3], const double dir[3],
                        double t_min, double t_max, double* t);
/* scatter (hitrecord.nim:82-132) with the counter RNG starting at draw *k (updated) */
int orc_ref64_scatter(const orc_sphere* s, const double in_orig[3], const double in_dir[3],
                      const double p[3], const double normal[3], int front_face,
                      const uint32_t key[2], uint32_t* k, double attenuation[3],
                      double out_orig[3], double out_dir[3]);
/* getRay (camera.nim:34-39) */
void orc_ref64_get_ray(const orc_camera* cam, double s, double t, const uint32_t key[2], uint32_t* k,
                       double orig[3], double dir[3]);
/* rayColor (raytracer.nim:5-20) of one caller-given ray; scattering draws from the counter stream of path
 * (pixel = index, sample 0) */
void orc_ref64_ray_color(const orc_sphere* spheres, int n, const double orig[3], const double dir[3], int depth,
                         uint64_t seed, uint32_t index, double colour[3]);
/* the same ray with the renderer's arithmetic (first query in binary64 over [t_min, t_max], later segments in
 * binary32): what rtow_trace_rays must return bit for bit.  Returns the list index hit first, -1 for none. */
int orc_mirror32_trace_ray(const orc_sphere* spheres, int n_spheres, const double orig[3], const double dir[3],
                           double t_min, double t_max, int max_depth, uint64_t seed, uint32_t index, double colour[3],
                           double* t);
/* scatter (hitrecord.nim:127-132) for one triple in the device's binary32 arithmetic (material = the
 * material fields of *material; its centre and radius are not used) */
int orc_mirror32_scatter(const orc_sphere* material, const double in_dir[3], const double normal[3], int front_face,
                         uint64_t seed, uint32_t index, double attenuation[3], double out_dir[3]);
/* writeColor quantisation of one channel sum (raytracer.nim:22-28) */
int orc_ref64_quantise(double channel_sum, int spp);

#ifdef __cplusplus
}
#endif
#endif
