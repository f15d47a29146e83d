// Launch interface between the C ABI layer and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "rtow_path.h"

namespace rtow {

constexpr int kTraceThreads = 256;   // 4 wavefronts per workgroup
constexpr int kGeomPad = 8;
constexpr int kMaxSpheres = (1 << 26) - 16;   // list indices travel in 26 bits beside a 6-bit lane number
constexpr int kPoolFields = 14;      // per-wave pool, dwords per lane: 64 scattered paths (o, d, throughput, stream, pixel, sphere)
                                     // or, during a primary scan, the 64 camera rays in binary64 (o, d, |d|^2)
constexpr int kMaxSpheresExactInLds = 512; // up to this many entries the exact-test records sit in LDS too (8 KB)
constexpr int kMaxSpheresKInLds = 2048;  // up to this many table entries the filter reads -k from an LDS copy of the whole table (8 KB)          // the regular sphere table is padded to a multiple of this

struct TraceParams {
  CameraParams64 cam;        // must stay first: the kernel reads it straight from its argument segment
  float slab_inv_rmin;       // 1 / (smallest group radius of the slab test): turns the test's absolute slack into a scale of the ray
  float pad_;
  const float* geom;         // n_geom_pad/4 + 2 groups of 16 floats (layout: scan_regular), list order; the last four of a group
                             // are k for tables beyond kMaxSpheresKInLds entries, -r^2 (slab test) for the others
  const float* geom_nk;      // -k per list entry (n_geom_pad + 8 values): copied to LDS by each workgroup
  const float4* geom_exact;  // (cx, cy, cz, r^2) per list entry, fetched per lane for the exact test
  const double4* geom64;     // (cx, cy, cz, r) per list entry in binary64: the exact test of primary rays
  const float* geom_slab;    // the slab test's radius per group of four (negative: never passes) for tables beyond kMaxSpheresKInLds entries
  const float* span_mid;     // interval test of the camera rays: per list entry the middle of the range of the image coordinate s
                             // in which a camera ray can meet the sphere (n_geom_pad + 4 kSpanPadGroups values), then the same for t ...
  const float* span_half;    // ... and per group of four the largest half width of its four ranges (negative: never passes;
                             // n_geom_pad / 4 + kSpanPadGroups values for s, then as many for t)
  const uint32_t* tile_mask; // mask test of the camera rays: per group of four one bit per tile of the image in which a camera ray can
                             // meet one of its spheres (n_geom_pad / 4 rounded up to a multiple of 16, + kTileMaskPad values)
  float tile_ks, tile_cs;    // column of the tile a ray with image coordinate s aims at: trunc(med3(s * tile_ks + tile_cs, 0, tile_smax))
  float tile_kt, tile_ct;    // its row, from t, likewise
  float tile_smax, tile_tmax;// columns - 1, rows - 1 (columns * rows <= 32)
  uint32_t tile_ns;          // columns: the tile's bit is column + tile_ns * row
  int n_geom_pad;
  int n_list;                // spheres in the list (= entries of geom64 and shade)
  float cmax;                // max(|c| + |r|) over the regular spheres: scales the filter margin
  const BigSphere* big;      // spheres intersected in binary64
  int n_big;
  const SphereShade* shade;  // one record per list entry
  int W, H, spp, max_depth;  // spp: samples per pixel traced by THIS launch
  uint32_t sample_base;      // index of its first sample (progressive passes continue a pixel's streams)
  uint32_t seed_lo, seed_hi;
  const int32_t* rows;       // image rows to render (0 = top), or nullptr for 0..n_rows-1
  int n_rows;
  // Work is handed out in runs ("chunks") of consecutive paths in pixel-major order, chunk_max
  // paths at a time while plenty is left and down to chunk_min towards the end of the frame, so
  // that the last waves to finish are at most one short chunk behind the others.
  uint32_t chunk_max, chunk_min;    // multiples of 64, chunk_min <= chunk_max
  uint32_t chunk_taper;             // a chunk is (paths left / chunk_taper) once that is below chunk_max; chunk_taper * chunk_max < 2^32
  uint32_t n_waves;                 // wavefronts in the grid
  uint32_t inv_spp;                 // ceil(2^32 / spp): n / spp == mulhi(n, inv_spp) while n * spp < 2^32
  unsigned long long total_paths;   // n_rows * W * spp
  unsigned long long* path_counter; // next path to hand out; zeroed before launch
  unsigned long long* accum;        // n_rows*W*3 fixed-point sums, zeroed before launch
  int32_t* hit_id;                  // n_rows*W or nullptr
  unsigned long long* seg_counter;  // zeroed before launch; [2..5] receive the first wave's clock stamps:
                                    // shader clock and constant 100 MHz clock at its start, the same at its end
};

// rayColor / hit for n caller-given rays, or getRay for n (s, t) pairs: one lane per ray
struct RayParams {
  CameraParams64 cam;
  const double* orig;        // n*3 ray origins          (getRay: n values of s)
  const double* dir;         // n*3 ray directions       (getRay: n values of t)
  int n;
  double t_min, t_max;       // interval of the first hit query (hittables.nim:79)
  int max_depth;             // rayColor's depth; <= 0: hit query only
  uint32_t seed_lo, seed_hi;
  const float* geom_nk; const float4* geom_exact; const double4* geom64;
  const BigSphere* big; int n_big;
  const SphereShade* shade;
  int n_list;
  float cmax;
  double* color;             // n*3 or nullptr           (getRay: n*3 origins out)
  int32_t* hit_id;           // n or nullptr
  double* t;                 // n or nullptr             (getRay: n*3 directions out)
};
void launch_rays(const RayParams& p, bool get_rays, hipStream_t stream);

// scatter (hitrecord.nim:127-132) for n caller-given (material, incoming ray, hit record) triples
struct ScatterParams {
  int n;
  uint32_t seed_lo, seed_hi;
  const double* dir;         // n*3 r_in.dir (not normalised)
  const double* normal;      // n*3 rec.normal (unit, already turned against the ray)
  const uint8_t* front;      // n   rec.frontFace
  const uint8_t* kind;       // n   material kind
  const double* albedo;      // n*3 (ignored for a dielectric)
  const double* fuzz;        // n
  const double* ir;          // n
  double* attenuation;       // n*3 out
  double* out_dir;           // n*3 out: scattered.dir
  uint8_t* ok;               // n   out: the proc's result
};
void launch_scatter(const ScatterParams& p, hipStream_t stream);

// slab_axis: what the camera rays see before the filter: the slab test along axis 0, 1 or 2 (see pair_slab), the
// bundle test (kBundleTest; see scan_bundle), the interval test (kIntervalTest; see scan_interval), the mask test in front of
// the interval test (kMaskTest; see scan_masked), or -1 for nothing
constexpr int kBundleTest = 3;
constexpr int kIntervalTest = 4;
constexpr int kMaskTest = 5;
constexpr int kTileMaskPad = 32;    // all-zero masks behind the last block of 16 groups (the mask test's loop requests ahead)
constexpr int kTileFrameMin = 16;   // the masks are made for frames at least this wide and high (s, t <= 1 + 1 / (kTileFrameMin - 1))
constexpr int kSpanPadGroups = 4;   // all-padding groups behind the last one in the interval test's tables (its loop requests two pairs ahead)
void launch_trace(const TraceParams& p, int slab_axis, int grid_blocks, hipStream_t stream);
void launch_resolve(const unsigned long long* accum, uint8_t* rgb, uint32_t n_channels, int spp,
                    hipStream_t stream);
int trace_blocks_per_cu(int n_geom_pad);
// mismatches[0..1] += inputs (of all 2^32) where the kernels' sqrt / reciprocal differ from sqrtf / 1.0f/x
void launch_arithmetic_selftest(unsigned long long* mismatches, hipStream_t stream);

// P3 text of an 8-bit image, one "r g b\n" line per pixel (raytracer.nim:28), formatted on the device.
constexpr int kPpmBlock = 256;   // pixels per workgroup
// pass 1: bytes of text per workgroup of pixels -> block_len[ceil(n_pixels / kPpmBlock)]
void launch_ppm_lengths(const uint8_t* rgb, uint32_t n_pixels, uint32_t* block_len, hipStream_t stream);
// pass 2: exclusive prefix sum of block_len -> block_off[n_blocks], grand total -> block_off[n_blocks]
void launch_ppm_scan(const uint32_t* block_len, uint32_t n_blocks, unsigned long long* block_off, hipStream_t stream);
// pass 3: the text itself, text[block_off[b] ...] for every workgroup b
void launch_ppm_text(const uint8_t* rgb, uint32_t n_pixels, const unsigned long long* block_off, char* text,
                     hipStream_t stream);

}  // namespace rtow
