/* rtow.h - C ABI of the MI355X (gfx950) renderer for the "Ray Tracing in One Weekend" hot path.
 *
 * The library replaces the body of the reference's
 *     proc render*(img: Image, f: string, world: var HittablesList, camera: Camera,
 *                  samplesPerPixel, maxDepth: int)                      (raytracer.nim:30-46)
 * i.e. the j/i/s loops and everything they call: getRay (camera.nim:34-39), rayColor
 * (raytracer.nim:5-20), HittablesList.hit (hittables.nim:79-88), Sphere.hit
 * (hitrecord.nim:55-79), scatter (hitrecord.nim:82-132) and the colour maths of writeColor
 * (raytracer.nim:22-28).  The host keeps the reference's types; it flattens them into the plain
 * arrays below, makes one render call, and writes the PPM text itself.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 on success or a negative rtow_status and
 *     leaves a message for rtow_last_error() (thread local);
 *   - the caller owns every host buffer; the library owns device state inside rtow_ctx;
 *   - one rtow_ctx per host thread and per GPU; a context is not thread safe;
 *   - the PRIMARY segment of every path - the camera ray (raytracer.nim:41-43, camera.nim:34-39), its
 *     closest hit (hittables.nim:79-88 over hitrecord.nim:55-71) and the sky colour of a miss
 *     (raytracer.nim:18-20) - is evaluated in binary64 with the reference's operations in the
 *     reference's order, from the binary64 scene and camera: primary hit ids, and the pixels of a
 *     one-sample depth-one render, are those of the reference arithmetic.  From the first hit point
 *     (rounded to binary32) on, all arithmetic is binary32, except the intersection of spheres with
 *     |radius| >= 100, which keeps a binary64 discriminant, and the final gamma/quantise step;
 *   - deviations from the reference in the binary32 segments, both to keep rounding noise out of the
 *     image: a ray scattered on sphere j can meet j again only at its far root (the reference would
 *     accept any root >= t_min, including a near root that rounding has pushed above it), and one
 *     sample's channel radiance is clamped to [0, 3.5] before it is summed in fixed point (never
 *     reached while albedos are <= 1, which rtow_set_scene enforces);
 *   - a pixel's channel sums are 64-bit integers in units of 2^-40.  A binary32 sample is cut to 30
 *     fractional bits before it is added, the binary64 sky colour of a camera ray to 40; writeColor
 *     (raytracer.nim:22-28) then runs in binary64 on the sum.  At one sample per pixel the 8-bit value
 *     is therefore the reference's unless a channel lies within 2^-40 above a quantisation boundary;
 *   - randomness: draw k of path (pixel, sample) is a pure function of (seed, pixel, sample, k),
 *     pixel = y*w + x with y = 0 the top image row.  It replaces the reference's single global
 *     std/random stream (raytracer.nim:41-42,135; basetypes.nim:112,132; hitrecord.nim:119),
 *     so an image does not depend on how it is split over GPUs.
 */
#ifndef RTOW_H
#define RTOW_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtow_ctx rtow_ctx;

typedef enum {
  RTOW_OK = 0,
  RTOW_ERR_INVALID = -1,  /* bad argument (the reference would hit an assert or a Nim exception) */
  RTOW_ERR_DEVICE = -2,   /* a HIP call failed; message carries hipGetErrorString */
  RTOW_ERR_STATE = -3,    /* render before set_scene/set_camera */
  RTOW_ERR_NOMEM = -4
} rtow_status;

/* MaterialKind, hitrecord.nim:16-17 */
enum { RTOW_LAMBERTIAN = 0, RTOW_METAL = 1, RTOW_DIELECTRIC = 2 };

/* HittableKind, hittables.nim:4-5: what an entry of the list is.  The reference's variant has the single
 * member htSphere; the value travels through the ABI so that a list with another member is refused by name
 * instead of being read as spheres. */
enum { RTOW_HT_SPHERE = 0 };

/* How rtow_render_multi brought the rows of its last frame to the host (rtow_stats.gather_path). */
enum {
  RTOW_GATHER_NONE = 0,            /* the context has not taken part in rtow_render_multi */
  RTOW_GATHER_HOST = 1,            /* every context's rows copied to the host in turn */
  RTOW_GATHER_RCCL = 2,            /* device to device over RCCL onto the first context's GPU, one copy to the host */
  RTOW_GATHER_HOST_AFTER_RCCL = 3  /* RCCL was asked for but could not be used or failed: rows copied to the host in turn */
};

/* What the camera rays of a context's last render saw in front of the sphere screen (rtow_stats.camera_screen).
 * None of them changes a hit, an image or a segment count; they differ in cost only. */
enum {
  RTOW_SCREEN_NONE = 0,      /* nothing: every camera ray puts every sphere through the screen (RTOW_NO_SLAB_TEST) */
  RTOW_SCREEN_SLAB = 1,      /* per (ray, sphere): distance between centre and the ray's line seen along one world axis */
  RTOW_SCREEN_INTERVAL = 2,  /* per (ray, sphere): the ray's image coordinates against the ranges in which the sphere can
                                be met at all (tables made by the library from scene and camera), for the camera
                                initCamera builds (camera.nim:12-31); other cameras get the slab test */
  RTOW_SCREEN_BUNDLE = 3,    /* per (wave, sphere): the list against a wave's 64 camera rays as one packet; only with
                                RTOW_CAMERA_SCREEN=bundle - packet culling, not brute force per ray */
  RTOW_SCREEN_MASK = 4       /* per (ray, group of four spheres): the tile of the image the ray aims at against the tiles
                                in which the group's spheres can be met (one 32-bit word per group, from the same ranges),
                                then RTOW_SCREEN_INTERVAL on the groups that pass; the default for frames of at least
                                16 x 16 pixels with a camera the ranges cover */
};

/* The eight private fields of Camera (camera.nim:4-10), rounded to binary32 by the host after
 * initCamera (camera.nim:12-31) has computed them in binary64. */
typedef struct {
  float origin[3];
  float lower_left[3];
  float horizontal[3];
  float vertical[3];
  float u[3], v[3], w[3];
  float lens_radius;
} rtow_camera;

/* The same fields in binary64, exactly as initCamera computes them.  Primary rays are generated and
 * intersected in binary64 (see the conventions above), so a host that has the binary64 camera - the
 * reference does - should pass it unrounded. */
typedef struct {
  double origin[3];
  double lower_left[3];
  double horizontal[3];
  double vertical[3];
  double u[3], v[3], w[3];
  double lens_radius;
} rtow_camera_f64;

/* statistics of the most recent render on a context */
typedef struct {
  uint64_t ray_segments;  /* world.hit queries (raytracer.nim:11): the numerator of Mrays/s */
  uint64_t paths;         /* pixels x samples */
  float trace_ms;         /* device time of the path-tracing kernel (HIP events) */
  float total_ms;         /* device time of clear + trace + resolve (HIP events) */
  int32_t n_spheres;
  int32_t n_big;          /* spheres routed through the binary64 test */
  float shader_clock_mhz; /* clock the GPU ran at during that kernel: shader-clock ticks per tick of the
                             constant 100 MHz counter, read by the kernel's first wave as it starts and retires */
  int32_t gather_path;    /* RTOW_GATHER_*: set on every context of the last rtow_render_multi call */
  int32_t camera_screen;  /* RTOW_SCREEN_*: what the camera rays of that render saw in front of the sphere screen */
  int32_t reserved_;
} rtow_stats;

/* Message of the last failing call on this thread ("" if none). */
const char* rtow_last_error(void);

/* Library version: major*10000 + minor*100 + patch. */
int rtow_version(void);

/* Number of visible HIP devices. */
int rtow_device_count(int* count);

/* Create / destroy a context bound to one device.  Replaces nothing in the reference (it has
 * no device); it is where the HIP stream and the workspace live. */
int rtow_create(int device, rtow_ctx** ctx);
int rtow_destroy(rtow_ctx* ctx);

/* Upload a HittablesList (hittables.nim:10-13) flattened to structure-of-arrays, in list order:
 * index i here is insertion index i there, which decides ties (hittables.nim:84-88).
 * center_* / radius: Sphere.center, Sphere.radius (hitrecord.nim:11-14; a negative radius is
 * legal and flips the normal, raytracer.nim:72).  kind: RTOW_* per sphere.  albedo_rgb: 3 floats
 * per sphere (ignored for dielectrics).  fuzz: Metal.fuzz.  ir: Dielectric.ir.
 * All pointers are host pointers; the data is copied.  n may be 0 (every ray sees the sky) and at
 * most 2^26 - 16 (RTOW_ERR_INVALID beyond). */
int rtow_set_scene(rtow_ctx* ctx, int n, const float* center_x, const float* center_y,
                   const float* center_z, const float* radius, const uint8_t* kind,
                   const float* albedo_rgb, const float* fuzz, const float* ir);

/* The same list from binary64 arrays (Sphere.center, Sphere.radius, Color, fuzz and ir are all Nim
 * `float` = binary64 in the reference, basetypes.nim:8).  Primary rays are tested against these
 * values as given; later segments use their binary32 roundings.  rtow_set_scene is this call on
 * values that are already binary32.  Both refuse (RTOW_ERR_INVALID) a radius that is zero or not
 * finite in binary32, a centre that is not finite, an unknown kind, and an albedo component outside
 * [0, 1] on a Lambertian or Metal sphere (the reference accepts those; the fixed-point radiance
 * sums here could not represent the result faithfully). */
int rtow_set_scene_f64(rtow_ctx* ctx, int n, const double* center_x, const double* center_y,
                       const double* center_z, const double* radius, const uint8_t* kind,
                       const double* albedo_rgb, const double* fuzz, const double* ir);

/* The same list as the reference holds it: an array of Hittable variants (hittables.nim:4-13), entry i being
 * Hittable(kind: hittable_kind[i], hSphere: the i-th sphere of the arrays, which are those of rtow_set_scene_f64).
 * hit*(h: Hittable, ...) (hittables.nim:75-77) dispatches on the kind and has one branch, htSphere; so does this
 * call: RTOW_HT_SPHERE is the only kind it accepts, any other value is RTOW_ERR_INVALID with the index and the
 * value in the message, and nothing is uploaded.  hittable_kind == NULL means every entry is a sphere. */
int rtow_set_hittables(rtow_ctx* ctx, int n, const uint8_t* hittable_kind, const double* center_x,
                       const double* center_y, const double* center_z, const double* radius,
                       const uint8_t* material_kind, const double* albedo_rgb, const double* fuzz, const double* ir);

/* Upload the camera (camera.nim:4-10): binary32 fields (widened exactly) or the binary64 ones.
 * Once scene and camera are both known - in either order, and again whenever one of them changes - the library works
 * out on the host, in O(n), the ranges of the image coordinates in which each sphere can be met by a camera ray (the
 * tables of RTOW_SCREEN_INTERVAL and RTOW_SCREEN_MASK; it waits for the context's stream first).  That construction covers the camera
 * initCamera builds (camera.nim:12-31: u, v orthonormal, horizontal and vertical along them, the focus plane
 * perpendicular to u x v, in binary64); with any other eight vectors - which are accepted as before - the camera
 * rays get RTOW_SCREEN_SLAB.  Images do not depend on which. */
int rtow_set_camera(rtow_ctx* ctx, const rtow_camera* cam);
int rtow_set_camera_f64(rtow_ctx* ctx, const rtow_camera_f64* cam);

/* render (raytracer.nim:30-46) for the image rows listed in rows[0..n_rows): row 0 is the top
 * row, the first one the reference writes.  rows == NULL means rows 0..h-1 (then n_rows must be
 * h).  Outputs are n_rows x w in rows[] order, host pointers, any may be NULL:
 *   rgb_out      3 bytes per pixel: the three integers writeColor prints (raytracer.nim:28)
 *   accum_out    3 uint64 per pixel: per-channel radiance sums, fixed point, 40 fractional bits
 *   hit_id_out   list index hit by the primary ray of sample 0, -1 for the sky (debug plane)
 * Blocks until the image is in host memory.  w,h >= 2; max_depth >= 1; 1 <= spp <= 4 194 304 (what the
 * 64-bit sums hold; the limit applies to all the samples of a pixel, passes of rtow_render_pass included). */
int rtow_render(rtow_ctx* ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                const int32_t* rows, int n_rows, uint8_t* rgb_out, uint64_t* accum_out,
                int32_t* hit_id_out);

/* Same work, but asynchronous and device resident: rgb_dev is a DEVICE pointer to n_rows*w*3
 * bytes (may be NULL: the image then stays in the context's workspace).  Work is enqueued on
 * the context's stream; rtow_sync() waits for it.  Used when the caller gathers tiles from
 * several GPUs itself, and by the benchmark (nothing crosses PCIe in the timed region). */
int rtow_render_async(rtow_ctx* ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                      const int32_t* rows, int n_rows, void* rgb_dev);
int rtow_sync(rtow_ctx* ctx);

/* Progressive refinement: one more pass of n_samples samples per pixel on top of what the context's
 * accumulators hold.  first_sample == 0 starts an image (like rtow_render_async); first_sample > 0
 * must equal the number of samples accumulated so far for the same w, h, rows and seed
 * (RTOW_ERR_STATE otherwise).  After every pass the 8-bit image (in rgb_dev, or in the context when
 * rgb_dev == NULL) is writeColor's (raytracer.nim:22-28) over all first_sample + n_samples samples.
 * Sample s of a pixel uses the same random stream whichever pass traces it and the sums are
 * integers, so passes of 100 + 400 samples leave exactly the image of one 500-sample render - a
 * caller can show or store intermediate frames of the reference's s loop (raytracer.nim:40) at no
 * cost in the final result.  Asynchronous; statistics describe the pass. */
int rtow_render_pass(rtow_ctx* ctx, int w, int h, int first_sample, int n_samples, int max_depth,
                     uint64_t seed, const int32_t* rows, int n_rows, void* rgb_dev);

/* rtow_render_async for callers that may not allocate between set-up and tear-down: tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference (which has no device memory to
 * manage).  The radiance sums and the work counters live in a fixed array inside the library's code object
 * instead of buffers from hipMalloc, and the call makes no hipMalloc / hipFree or other allocating call;
 * scene and camera must have been uploaded before (rtow_set_scene allocates).  It is NOT safe under HIP
 * graph capture: like every render call it waits for the context's previous frame and records timing events.
 * Differences from rtow_render_async:
 *   - rows_dev is a DEVICE pointer to n_rows row indices in [0, h) (or NULL with n_rows == h); the
 *     values are the caller's responsibility, the library cannot inspect them without a copy;
 *   - rgb_dev (device, n_rows*w*3 bytes) must not be NULL;
 *   - n_rows * w may not exceed 1 048 576 pixels (RTOW_ERR_INVALID beyond; 1200 x 800 fits), and
 *     1 <= spp <= 4 194 304 as for rtow_render;
 *   - the image cannot be continued by rtow_render_pass or fetched with rtow_read_image /
 *     rtow_format_ppm: nothing of it stays in the context.
 * The array (25 MB) exists once per GPU and is shared by every context on that GPU, so one context at a time
 * may have such a frame in flight there: the call takes the array for its context and gives it back when that
 * context is next synchronised (rtow_sync, or any call that waits for the frame) or destroyed; a call on
 * another context of the same GPU in between returns RTOW_ERR_STATE and renders nothing.  Calls on one context
 * follow each other on its stream.  The pixels equal rtow_render_async's bit for bit. */
int rtow_render_noalloc(rtow_ctx* ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                        const int32_t* rows_dev, int n_rows, void* rgb_dev);

/* Copy the image of the most recent render out of the context (waits for it): rgb_out n_rows*w*3
 * bytes, accum_out n_rows*w*3 uint64 fixed-point sums; either may be NULL. */
int rtow_read_image(rtow_ctx* ctx, uint8_t* rgb_out, uint64_t* accum_out);

/* The reference's per-ray procs, batched: for n caller-given rays (binary64, n*3 origins and n*3
 * directions, directions not normalised - camera.nim:38-39 does not normalise either)
 *   hit_id_out[i], t_out[i]   hit*(world, r, t_min, t_max, rec) (hittables.nim:79-88 over Sphere.hit,
 *                             hitrecord.nim:55-79): list index of the closest hit in [t_min, t_max]
 *                             and its t in units of |direction| (-1 and +inf for a miss); the rest of
 *                             the HitRecord follows from them on the host (p = orig + t*dir, ...);
 *   color_out[3*i..]          rayColor*(r, world, max_depth) (raytracer.nim:5-20), un-scaled radiance.
 * The arithmetic is the renderer's: this query in binary64 with the reference's operations, any further
 * bounce in binary32; ray i scatters with the random stream of path (pixel i, sample 0) of `seed`.
 * rayColor itself uses t_min = 0.001 and t_max = +inf (raytracer.nim:11); other values give the colour
 * of a ray whose FIRST query is restricted to [t_min, t_max].  max_depth = 0 answers the hit query
 * only (colours are 0, as rayColor returns at depth <= 0).  Any output may be NULL.  Host pointers;
 * blocks until the results are there.  Needs rtow_set_scene; n at most 2^27. */
int rtow_trace_rays(rtow_ctx* ctx, int n, const double* origins, const double* directions, double t_min,
                    double t_max, int max_depth, uint64_t seed, double* color_out, int32_t* hit_id_out,
                    double* t_out);

/* getRay*(c, s, t) (camera.nim:34-39) for n (s, t) pairs in binary64: origins_out and directions_out
 * (n*3 each) of the thin-lens rays.  Ray i takes its lens sample (randomInUnitDisk, basetypes.nim:130-134)
 * from the first draws of the stream of path (pixel i, sample 0) of `seed`, where the reference draws
 * from its one global stream.  Needs rtow_set_camera. */
int rtow_get_rays(rtow_ctx* ctx, int n, const double* s, const double* t, uint64_t seed,
                  double* origins_out, double* directions_out);

/* scatter*(m, r_in, rec, attenuation, scattered): bool (hitrecord.nim:127-132 over :82-125) for n independent
 * (material, incoming ray, hit record) triples.  In: directions (n*3, r_in.dir, not normalised), normals
 * (n*3, rec.normal: unit, already turned against the ray by setFaceNormal, hitrecord.nim:51-53),
 * front_face (n, rec.frontFace), and the material as rtow_set_scene takes it (kind, albedo_rgb n*3 - ignored
 * for a dielectric - fuzz, ir).  Out: attenuation_out (n*3), scattered_dir_out (n*3, scattered.dir, not
 * normalised; scattered.orig is rec.p, which the caller has) and scattered_out (n, the proc's result; 0
 * only for a Metal whose fuzzed reflection points below the surface, hitrecord.nim:99 - direction and
 * attenuation are delivered all the same, as the reference's var parameters are).
 * The arithmetic is the renderer's scatter: binary32 on the rounded inputs, with the unit direction of
 * r_in (the reference normalises it where Metal and Dielectric use it, hitrecord.nim:96,112).  Triple i
 * draws from the stream of path (pixel i, sample 0) of `seed`, from that stream's first draw: three per
 * trip of randomInUnitSphere (basetypes.nim:114-118) for Lambertian and Metal, one for a Dielectric that
 * can refract (hitrecord.nim:119).  Needs neither scene nor camera.  Host pointers; blocks until the
 * results are there; n at most 2^27. */
int rtow_scatter_rays(rtow_ctx* ctx, int n, const double* directions, const double* normals,
                      const uint8_t* front_face, const uint8_t* kind, const double* albedo_rgb,
                      const double* fuzz, const double* ir, uint64_t seed, double* attenuation_out,
                      double* scattered_dir_out, uint8_t* scattered_out);

/* One frame on several contexts at once - normally one context per GPU of this process, each
 * holding the same scene and camera.  The rows of the image - the outer j loop of render
 * (raytracer.nim:36) - are dealt to the contexts in interleaved tiles of tile_rows rows (tile t goes
 * to context t mod n_ctx; rtow_multi_plan below is that dealing), every context renders its rows as
 * rtow_render_async does, and the 8-bit rows are copied into rgb_out (host, h*w*3 bytes, top row first).
 * The image is byte-identical to what rtow_render produces on one context.  By default every context's rows
 * are copied to the host in turn.  With RTOW_MULTI_GATHER=rccl in the environment and the contexts on
 * pairwise different GPUs the rows are gathered device to device over RCCL (xGMI) onto the first context's
 * GPU - librccl.so is opened and the communicators are built BEFORE the frame is enqueued, all sends and
 * receives are enqueued before anything is waited for - and reach the host in one copy; if RCCL cannot be
 * loaded or set up, or the exchange fails, the rows are copied to the host in turn after all
 * (rtow_stats.gather_path says which way a frame went).  The RCCL branch with two or more GPUs has not run
 * on hardware yet (the build's machines have one GPU), which is why it is opt-in.  ray_segments_out (may be
 * NULL) receives the total number of world.hit queries.  Each context's rtow_get_stats then describes its own
 * share.
 * Hosts that run one PROCESS per GPU use rtow_render_async with their own row list instead and
 * exchange the rows themselves (the Python host does so with torch.distributed over RCCL). */
int rtow_render_multi(rtow_ctx* const* ctxs, int n_ctx, int w, int h, int spp, int max_depth,
                      uint64_t seed, int tile_rows, uint8_t* rgb_out, uint64_t* ray_segments_out);

/* The dealing of rows behind rtow_render_multi, as plain arithmetic (no device is touched): image row y
 * (0 = top) belongs to context (y / tile_rows) mod n_ctx and is that context's index_in_ctx[y]-th row; context k
 * renders rows_per_ctx[k] rows (0 when there are fewer tiles than contexts).  *slot_bytes is what each
 * context's rows occupy in the gather buffer of the RCCL path: the largest share, 3*w bytes per row, rounded
 * up to a multiple of 256; row y of the image is read from byte  slot * ctx_of_row[y] + 3*w * index_in_ctx[y].
 * ctx_of_row and index_in_ctx have h entries, rows_per_ctx n_ctx; any output may be NULL.  Hosts that run one
 * process per GPU and exchange the rows themselves deal the rows the same way. */
int rtow_multi_plan(int w, int h, int n_ctx, int tile_rows, int32_t* ctx_of_row, int32_t* index_in_ctx,
                    int32_t* rows_per_ctx, uint64_t* slot_bytes);

/* The ranges behind RTOW_SCREEN_INTERVAL as plain arithmetic (no device is touched; nothing in the reference
 * corresponds to it): for a camera and n spheres, mid_s[i] +- half_s[i] is the range of getRay's first argument s
 * (camera.nim:34) outside which no camera ray - whatever its t and its point on the lens - can meet sphere i, and
 * mid_t[i] +- half_t[i] the same for t; widened for the renderer's binary32 comparison |mid - s| <= half.  half = +inf:
 * any value (the sphere reaches across the lens plane); half < 0: no camera ray can meet the sphere (it lies wholly
 * behind the lens) or the sphere is not part of the test (|radius| >= 100: every ray tests those anyway).
 * *covered = 0 when the camera is not one the construction covers (see rtow_set_camera_f64); the arrays are then left
 * alone.  The renderer compares against the largest half width of a sphere's group of four list entries.  For hosts
 * that want the bounds for their own use, and for tests.  Any output may be NULL. */
int rtow_interval_plan(const rtow_camera_f64* cam, int n, const double* center_x, const double* center_y,
                       const double* center_z, const double* radius, float* mid_s, float* half_s, float* mid_t,
                       float* half_t, int* covered);

/* The words behind RTOW_SCREEN_MASK as plain arithmetic (no device is touched; nothing in the reference corresponds
 * to it).  The image is cut into *columns x (32 / *columns) tiles; a ray with image coordinates (s, t) - getRay's
 * arguments, camera.nim:34 - aims at column trunc(med3(s * grid[0] + grid[1], 0, grid[4])) and row
 * trunc(med3(t * grid[2] + grid[3], 0, grid[5])), all in binary32 with one fused multiply-add, and its bit is
 * column + *columns * row.  group_mask[g], g < (n + 3) / 4, has the bits of the tiles in which a camera ray can meet
 * one of the spheres 4g .. 4g+3: a ray whose bit is not set there meets none of them.  Valid for frames at least 16
 * pixels wide and high (s, t <= 1 + 1/15).  *covered as in rtow_interval_plan.  Any output may be NULL. */
int rtow_mask_plan(const rtow_camera_f64* cam, int n, const double* center_x, const double* center_y,
                   const double* center_z, const double* radius, uint32_t* group_mask, float* grid, int* columns,
                   int* covered);

/* The PPM file of the most recent render on this context, as the reference writes it: the header
 * "P3\n{w} {rows}\n255\n" (raytracer.nim:35) followed by one "r g b\n" line per pixel
 * (writeColor, raytracer.nim:28), rows in the order they were rendered.  The text is composed on the
 * device (about 11 bytes per pixel, 100 MB at 3840x2160) and copied into out[0..*length).
 * binary != 0 gives the P6 variant instead (same header with P6, then 3 raw bytes per pixel).
 * out == NULL only reports *length.  capacity < *length is RTOW_ERR_INVALID (with *length set).
 * Needs a render whose image stayed in the context (rtow_render, or rtow_render_async /
 * rtow_render_multi with rgb_dev == NULL); waits for it if it is still running. */
int rtow_format_ppm(rtow_ctx* ctx, int binary, char* out, uint64_t capacity, uint64_t* length);

/* Self test of the device arithmetic: the kernels take square roots and reciprocals with short
 * correctly-rounded sequences of their own; this runs every one of the 2^32 binary32 bit patterns
 * through them and through the compiler's sqrtf(x) and 1.0f / x on the GPU and returns how many
 * results differ (mismatches[0] sqrt, [1] reciprocal; both must be 0 - the CPU restatement the
 * renderer is checked against uses the C library's correctly rounded operations).  About 0.1 s. */
int rtow_selftest_arithmetic(rtow_ctx* ctx, uint64_t mismatches[2]);

/* Statistics of the last completed render (call after rtow_render or rtow_sync). */
int rtow_get_stats(rtow_ctx* ctx, rtow_stats* out);

/* The context's HIP stream (hipStream_t) as an opaque pointer, for callers that order their own
 * device work after a render. */
int rtow_get_stream(rtow_ctx* ctx, void** stream);

#ifdef __cplusplus
}
#endif
#endif /* RTOW_H */
