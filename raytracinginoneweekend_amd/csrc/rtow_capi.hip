// C ABI of the renderer (declared in include/rtow.h): context, scene/camera upload, render.
// No CPU rendering path exists here: every entry point that produces pixels launches the HIP
// kernels and fails with RTOW_ERR_DEVICE if it cannot.
#include "../../include/rtow.h"
#include "rtow_kernels.h"

#include <algorithm>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <rccl/rccl.h>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(RTOW_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                  __FILE__, __LINE__);                                                      \
  } while (0)

// a device buffer that only grows
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Work memory of rtow_render_noalloc: a fixed array inside the code object instead of buffers from
// hipMalloc.  Every GPU the library runs on holds its own copy (25 MB) until the process ends, and
// all contexts on one GPU share it.  Layout: 256 bytes of counters (segment count, next path), then
// three 64-bit radiance sums per pixel.
constexpr size_t kCounterBytes = 48;   // per frame: segment count, next path, four clock stamps (8 bytes each)
constexpr uint32_t kNoAllocMaxPixels = 1u << 20;   // 1 048 576: a 1200 x 800 frame fits
constexpr size_t kScratchHeader = 256;
constexpr size_t kScratchBytes = kScratchHeader + sizeof(unsigned long long) * 3 * (size_t)kNoAllocMaxPixels;
constexpr int kMaxDevices = 64;

}  // namespace

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

namespace {
// Device address of `scratch` on each GPU (the array's name is only a host-side handle): looked up
// once per device by rtow_create with that device current.  The same value whichever thread stores it.
std::atomic<unsigned char*> g_scratch_ptr[kMaxDevices];
// The context whose rtow_render_noalloc frame currently owns this GPU's copy of the array (nullptr: free).  Taken by
// the call, given back when that context is synchronised (rtow_sync and every call that waits for the frame) or
// destroyed: a second context that asks for the array in between is refused instead of racing on it.
std::atomic<rtow_ctx*> g_scratch_owner[kMaxDevices];
}  // namespace

// RCCL, bound at run time: the library is only needed by rtow_render_multi on two or more GPUs, so it is
// opened on first use (the copy already in the process - a host such as PyTorch brings its own - or the
// system one) instead of being a load-time dependency of every user of this library.
namespace {
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::map<std::vector<int>, std::vector<ncclComm_t>> comms;   // one clique per ordered device list, kept for the process
};
Rccl g_rccl;
std::mutex g_rccl_mutex;   // guards g_rccl: communicator creation is rare, collectives on one clique are serialised

// Communicators are destroyed when the process exits, before the libraries under them unload (a handler
// registered now runs ahead of the destructors of everything loaded before this point).
void rccl_shutdown() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  for (auto& clique : g_rccl.comms)
    for (ncclComm_t comm : clique.second)
      if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
  g_rccl.comms.clear();
}

// nullptr on success, otherwise what is missing
const char* rccl_load() {
  if (g_rccl.handle) return nullptr;
  // Order: a copy already in the process (a second, different librccl beside it ends badly: both answer to the
  // soname librccl.so.1, and whoever is loaded first is handed to the other's users), then the file the host
  // names in RTOW_RCCL_LIB (the Python package points it at PyTorch's bundled copy, so that a later
  // `import torch` finds the library it was built with), then the system's.  Local scope: only the entry
  // points looked up below are used, and RCCL's other symbols must not interpose on libraries loaded later.
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!h) {
    const char* named = getenv("RTOW_RCCL_LIB");
    if (named && *named) h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    if (h) break;
    h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return "librccl.so not found";
  Rccl r;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
  r.Send = (decltype(r.Send))dlsym(h, "ncclSend");
  r.Recv = (decltype(r.Recv))dlsym(h, "ncclRecv");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString)
    return "librccl.so lacks a symbol";
  r.handle = h;
  r.comms.swap(g_rccl.comms);
  g_rccl = std::move(r);
  atexit(rccl_shutdown);
  return nullptr;
}
}  // namespace

// Grid of image tiles of the mask test (rtow_kernels.hip, scan_masked): a ray with image coordinates (s, t) aims at
// column trunc(med3(s * ks + cs, 0, smax)) and row trunc(med3(t * kt + ct, 0, tmax)).
struct TilePlan {
  float ks = 0.0f, cs = 0.0f, kt = 0.0f, ct = 0.0f, smax = 0.0f, tmax = 0.0f;
  uint32_t ns = 1;   // columns; rows = tmax + 1; columns * rows <= 32
};

struct rtow_ctx {
  int device = 0;
  int n_cu = 0;
  int blocks_per_cu = 1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // start, after clear, after trace, end

  // scene
  bool have_scene = false, have_camera = false;
  int n_spheres = 0, n_geom_pad = 0, n_big = 0;
  DevBuf geom, geom_rr, geom_nk, geom_slab, geom64, big, shade;
  float cmax = 0.0f;
  float slab_inv_rmin = 1.0f;                 // 1 / (smallest group radius of the slab test), see set_scene_impl
  int slab_axis = 1;                          // axis of the camera rays' slab test, chosen from the camera
  // interval test of the camera rays (refresh_spans): tables derived from scene AND camera
  DevBuf span_mid, span_half;                 // [0]: ranges of s (across the image), [1]: ranges of t (up the image), back to back
  std::vector<double4> host_geom64;           // the scene as given (centre, radius), kept for refresh_spans
  std::vector<uint8_t> host_regular;          // per list entry: 1 = in the binary32 table (|radius| < kBigRadius)
  std::vector<float> host_span_mid, host_span_half;   // staging of the uploads (must outlive the copies); s ranges, then t ranges
  bool spans_valid = false;                   // the tables match the current scene and camera, and the camera is a regular one
  // mask test of the camera rays (build_tile_masks): derived from the same ranges
  DevBuf tile_mask;
  std::vector<uint32_t> host_tile_mask;       // staging of the upload
  TilePlan tiles;
  int screen_launched = 0;                    // RTOW_SCREEN_* of the frame in flight
  rtow::CameraParams64 cam;

  // per-render workspace
  DevBuf ppm_len, ppm_off, ppm_text;          // device-side PPM text of the last image (rtow_format_ppm)
  uint64_t ppm_bytes = 0;                     // length of that text, 0 = not composed yet
  bool image_in_workspace = false;            // the last render left its 8-bit image in `rgb`
  bool sums_in_workspace = false;             // ... and its radiance sums in `accum`
  DevBuf ray_in, ray_out;                     // staging of rtow_trace_rays / rtow_get_rays
  DevBuf gather;                              // rtow_render_multi over RCCL: the other GPUs' rows land here (context 0)
  uint8_t* gather_host = nullptr;             // pinned copy of it
  size_t gather_host_cap = 0;
  DevBuf rows, accum, hit_id, rgb, counters;  // counters: segment count (u64), path counter (u64), four clock stamps (u64)
  unsigned char* scratch_dev = nullptr;       // this GPU's copy of the no-allocation work memory
  uint64_t* host_counters = nullptr;          // pinned, kCounterBytes
  char* staging[2] = {nullptr, nullptr};      // pinned, kStagingBytes each: large device-to-host copies go through them
  hipEvent_t staging_ev[2] = {nullptr, nullptr};

  // last render
  bool pending = false;
  bool holds_scratch = false;                 // the frame in flight keeps its sums in this GPU's no-allocation work memory
  int last_w = 0, last_h = 0, last_spp = 0, last_rows = 0;
  int samples_done = 0;                       // samples per pixel in the accumulators
  uint64_t last_seed = 0;
  std::vector<int32_t> last_row_list;         // rows[] of the render the accumulators belong to (empty = all rows)
  bool last_row_list_valid = false;
  rtow_stats stats = {};
};

// Gives this GPU's no-allocation work memory back once the frame that used it has been waited for.
static void release_scratch(rtow_ctx* c) {
  if (!c->holds_scratch) return;
  c->holds_scratch = false;
  rtow_ctx* expected = c;
  if (c->device < kMaxDevices) g_scratch_owner[c->device].compare_exchange_strong(expected, nullptr, std::memory_order_acq_rel);
}

// Device-to-host copy of a large buffer into ordinary (pageable) caller memory: the runtime stages
// such copies through a small internal buffer at a few GB/s; two pinned 8 MB buffers used in turn -
// one being filled by the copy engine while the host empties the other - run at PCIe rate.
constexpr size_t kStagingBytes = 8u << 20;
static int copy_to_host(rtow_ctx* c, void* dst, const void* src_dev, size_t bytes) {
  // Every copy, small ones included, runs on the context's own stream through the context's own pinned buffers:
  // a blocking hipMemcpy would go through the NULL stream and the runtime's shared staging memory, which the
  // context's (non-blocking) stream is not ordered against and which all host threads of the process share.
  for (int k = 0; k < 2; ++k) {
    if (!c->staging[k]) HIP_TRY(hipHostMalloc((void**)&c->staging[k], kStagingBytes, hipHostMallocDefault));
    if (!c->staging_ev[k]) HIP_TRY(hipEventCreateWithFlags(&c->staging_ev[k], hipEventDisableTiming));
  }
  const size_t n_chunks = (bytes + kStagingBytes - 1) / kStagingBytes;
  auto chunk_bytes = [&](size_t i) { return i + 1 < n_chunks ? kStagingBytes : bytes - i * kStagingBytes; };
  auto issue = [&](size_t i) -> hipError_t {
    hipError_t e = hipMemcpyAsync(c->staging[i & 1], (const char*)src_dev + i * kStagingBytes, chunk_bytes(i),
                                  hipMemcpyDeviceToHost, c->stream);
    return e == hipSuccess ? hipEventRecord(c->staging_ev[i & 1], c->stream) : e;
  };
  HIP_TRY(issue(0));
  for (size_t i = 0; i < n_chunks; ++i) {
    if (i + 1 < n_chunks) HIP_TRY(issue(i + 1));
    HIP_TRY(hipEventSynchronize(c->staging_ev[i & 1]));
    memcpy((char*)dst + i * kStagingBytes, c->staging[i & 1], chunk_bytes(i));
  }
  return RTOW_OK;
}

extern "C" {

const char* rtow_last_error(void) { return g_err.c_str(); }

int rtow_version(void) { return 10200; }   // 1.1.0: rtow_stats grew by camera_screen; 1.2.0: rtow_mask_plan, RTOW_SCREEN_MASK

int rtow_device_count(int* count) {
  if (!count) return fail(RTOW_ERR_INVALID, "rtow_device_count: count is NULL");
  HIP_TRY(hipGetDeviceCount(count));
  return RTOW_OK;
}

int rtow_create(int device, rtow_ctx** out) {
  if (!out) return fail(RTOW_ERR_INVALID, "rtow_create: ctx is NULL");
  *out = nullptr;
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return fail(RTOW_ERR_INVALID, "rtow_create: device %d of %d", device, n);
  HIP_TRY(hipSetDevice(device));
  rtow_ctx* c = new (std::nothrow) rtow_ctx;
  if (!c) return fail(RTOW_ERR_NOMEM, "rtow_create: out of host memory");
  c->device = device;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->host_counters, kCounterBytes, hipHostMallocDefault);
  if (e == hipSuccess) e = c->counters.reserve(kCounterBytes);
  if (e == hipSuccess && device < kMaxDevices) {
    unsigned char* p = g_scratch_ptr[device].load(std::memory_order_acquire);
    if (!p) {
      e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));   // `device` is current (hipSetDevice above)
      if (e == hipSuccess) g_scratch_ptr[device].store(p, std::memory_order_release);
    }
    c->scratch_dev = p;
  }
  if (e != hipSuccess) {
    rtow_destroy(c);
    return fail(RTOW_ERR_DEVICE, "rtow_create: %s", hipGetErrorString(e));
  }
  c->n_cu = prop.multiProcessorCount;
  *out = c;
  return RTOW_OK;
}

int rtow_destroy(rtow_ctx* c) {
  if (!c) return RTOW_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  release_scratch(c);
  c->geom.release(); c->geom_rr.release(); c->geom_nk.release(); c->geom_slab.release(); c->span_mid.release(); c->span_half.release(); c->tile_mask.release(); c->geom64.release(); c->big.release(); c->shade.release(); c->rows.release(); c->accum.release();
  c->hit_id.release(); c->rgb.release(); c->counters.release();
  c->ppm_len.release(); c->ppm_off.release(); c->ppm_text.release(); c->ray_in.release(); c->ray_out.release(); c->gather.release();
  if (c->gather_host) (void)hipHostFree(c->gather_host);
  if (c->host_counters) (void)hipHostFree(c->host_counters);
  for (int k = 0; k < 2; ++k) {
    if (c->staging[k]) (void)hipHostFree(c->staging[k]);
    if (c->staging_ev[k]) (void)hipEventDestroy(c->staging_ev[k]);
  }
  for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return RTOW_OK;
}

}  // extern "C"

// Interval test of the camera rays (rtow_kernels.hip, scan_interval).  A camera ray is the line from a point of the
// lens disk to the point P(s, t) = llc + s hor + t ver of the focus plane (camera.nim:34-39).  Seen along the camera's
// up axis v - dropping that coordinate - lens and focus plane are two parallel lines a distance f apart, the sphere is
// a circle of the same radius, and a ray that meets the sphere projects to a line that meets the circle.  For a lens
// point x_l and a circle (X, Z, rho) wholly in front of the lens line (Z > rho), the lines through x_l that meet the
// circle reach the focus line between x_l + f m_lo and x_l + f m_hi,
//     m = (A Z -+ rho sqrt(A^2 + Z^2 - rho^2)) / (Z^2 - rho^2),   A = X - x_l,
// (the two tangents); m_hi is convex and m_lo concave in x_l, so over the lens |x_l| <= R the extremes sit at its two
// ends.  That gives, per sphere, a range [s_lo, s_hi] of the image coordinate s outside which no camera ray can meet
// it; a sphere wholly behind the lens line is met by no ray, one that reaches across it gets the whole axis.  The
// kernel compares |s - middle| of every (ray, sphere) pair with the largest half width of the sphere's group of four:
// one packed subtraction per two spheres.  Ranges are widened by 2^-19 (|s_lo| + |s_hi| + 2): the subtraction runs in
// binary32 on roundings of s and of the middle.  The same construction with u in the place of v bounds t, the
// coordinate up the image: groups that pass for s are tested for t before they go on.  Everything here is evaluated
// in binary64 on the scene as given, which is what the binary64 test that decides a camera ray's hit sees.
// The construction needs the camera the reference builds (camera.nim:12-31): u, v orthonormal, hor and ver along
// them, the focus plane perpendicular to u x v and in front of the lens.  Any other set of eight vectors the ABI
// accepts leaves spans_valid false, and the camera rays see the slab test instead.
// The camera as the construction needs it, or covered = false.
struct SpanCamera {
  bool covered = false;
  double hu = 0, hv = 0, f = 0, R = 0;   // lengths of the focus plane's edges, focus distance, lens radius
  double w[3] = {0, 0, 0};               // u x v: the direction the camera looks away from
};

static double dot3d(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

static SpanCamera span_camera(const rtow::CameraParams64& cam) {
  SpanCamera s;
  s.hu = std::sqrt(dot3d(cam.hor, cam.hor));
  s.hv = std::sqrt(dot3d(cam.ver, cam.ver));
  s.w[0] = cam.u[1] * cam.v[2] - cam.u[2] * cam.v[1];
  s.w[1] = cam.u[2] * cam.v[0] - cam.u[0] * cam.v[2];
  s.w[2] = cam.u[0] * cam.v[1] - cam.u[1] * cam.v[0];
  double F[3];
  for (int k = 0; k < 3; ++k) F[k] = cam.origin[k] - 0.5 * cam.hor[k] - 0.5 * cam.ver[k] - cam.llc[k];   // = f w
  s.f = dot3d(F, s.w);
  s.R = cam.lens_radius;
  bool regular = std::isfinite(s.hu) && std::isfinite(s.hv) && s.hu > 0.0 && s.hv > 0.0 && std::isfinite(s.f) && s.f > 0.0 &&
                 std::isfinite(s.R) && s.R >= 0.0 && std::fabs(dot3d(cam.u, cam.u) - 1.0) < 1e-9 &&
                 std::fabs(dot3d(cam.v, cam.v) - 1.0) < 1e-9 && std::fabs(dot3d(cam.u, cam.v)) < 1e-9;
  for (int k = 0; k < 3 && regular; ++k)
    regular = std::fabs(cam.hor[k] - s.hu * cam.u[k]) <= 1e-9 * s.hu && std::fabs(cam.ver[k] - s.hv * cam.v[k]) <= 1e-9 * s.hv &&
              std::fabs(F[k] - s.f * s.w[k]) <= 1e-9 * (s.f + s.hu + s.hv) && std::isfinite(cam.origin[k]);
  s.covered = regular;
  return s;
}

// One sphere's range along one image axis (0: s, across the image; 1: t, up the image) as middle and half width;
// half = +inf: any value, half < 0: no camera ray can meet the sphere.
constexpr float kSpanNever = 1e30f;   // the middle of a sphere no ray can meet (and of the padding)
static void span_of_sphere(const rtow::CameraParams64& cam, const SpanCamera& sc, int axis, double4 sp, float& mid, float& half) {
  const double* across = axis == 0 ? cam.u : cam.v;   // the image axis the range is measured along
  const double extent = axis == 0 ? sc.hu : sc.hv;   // length of the focus plane's edge along it
  const double Rl = sc.R * (1.0 + 1e-9);
  const double rel[3] = {sp.x - cam.origin[0], sp.y - cam.origin[1], sp.z - cam.origin[2]};
  const double X = dot3d(rel, across), Z = -dot3d(rel, sc.w);
  // The camera's vectors may be off the ideal frame by 1e-9 of their lengths (span_camera), and the ray and this
  // arithmetic round in binary64: lens and focus points then sit within 1e-9 (hu + hv + f + R) of where the
  // construction puts them, which moves a line by up to (1 + Z / f) times that at the sphere's depth.
  const double rho = std::fabs(sp.w) * (1.0 + 1e-9) + 4e-9 * (sc.hu + sc.hv + sc.f + Rl + std::fabs(X) + std::fabs(Z)) * (1.0 + std::fabs(Z) / sc.f);
  mid = kSpanNever; half = -1.0f;
  if (Z < -rho) return;                        // wholly behind the lens line: never
  mid = 0.0f; half = INFINITY;
  if (!(Z > rho * (1.0 + 1e-6))) return;       // reaches across the lens line: any value
  const double k = Z * Z - rho * rho;
  double s_lo = INFINITY, s_hi = -INFINITY;
  for (int e = 0; e < 2; ++e) {
    const double xl = e ? Rl : -Rl, A = X - xl, root = rho * std::sqrt(A * A + k);
    s_lo = std::min(s_lo, (xl + sc.f * (A * Z - root) / k) / extent + 0.5);
    s_hi = std::max(s_hi, (xl + sc.f * (A * Z + root) / k) / extent + 0.5);
  }
  if (!std::isfinite(s_lo) || !std::isfinite(s_hi)) return;
  const double pad = 0x1p-19 * (std::fabs(s_lo) + std::fabs(s_hi) + 2.0);
  const float m = (float)(0.5 * (s_lo + s_hi));
  if (!std::isfinite(m) || std::fabs(m) >= kSpanNever) return;
  mid = m;
  // half width against the ROUNDED middle, rounded up
  half = std::nextafter((float)(std::max(s_hi - (double)m, (double)m - s_lo) + pad), INFINITY);
}

// Column (or row) of the tile a coordinate falls into: the kernel's tile_bit, operation for operation (one fused
// multiply-add in binary32, v_med3_f32 - which returns the smallest operand when one is NaN, here 0 - and a truncation).
// Monotonic in v for k >= 0, which is all the mask test needs from it.
static uint32_t tile_index(float v, float k, float c, float vmax) {
  const float x = std::fmaf(v, k, c);
  if (!(x > 0.0f)) return 0u;
  return (uint32_t)(x > vmax ? vmax : x);
}

// Mask test of the camera rays: per group of four spheres the tiles of the image its spheres' ranges (mid -+ half per
// list entry and axis; half < 0 = no camera ray can meet the sphere) reach into.  A ray that can meet a sphere has its
// binary32 s between the sphere's range ends rounded outwards, so - tile_index being monotonic - its column lies
// between the columns of those two values; the same for t and rows.  Image coordinates run from 0 to
// 1 + 1 / (kTileFrameMin - 1) for the frames the test is used on: a sphere whose range ends below 0 or starts beyond
// that on either axis gets no tile.  The grid covers the part of the image the other spheres' ranges span (what lies
// outside falls into the edge tiles, on both sides of the comparison), with the split into columns and rows that lets
// the fewest (tile, group) pairs through, weighted by tile area.
static void build_tile_masks(int n, const float* mid_s, const float* half_s, const float* mid_t, const float* half_t,
                             TilePlan& plan, std::vector<uint32_t>& masks) {
  const double top = 1.0 + 1.0 / (rtow::kTileFrameMin - 1);
  struct Box { float lo[2], hi[2]; };
  std::vector<Box> boxes((size_t)n);
  std::vector<uint8_t> used((size_t)n, 0);
  double h0[2] = {INFINITY, INFINITY}, h1[2] = {-INFINITY, -INFINITY};
  for (int j = 0; j < n; ++j) {
    const float mid[2] = {mid_s[j], mid_t[j]}, half[2] = {half_s[j], half_t[j]};
    bool in = true;
    Box b;
    for (int a = 0; a < 2 && in; ++a) {
      if (!(half[a] >= 0.0f)) { in = false; break; }
      const double lo = (double)mid[a] - (double)half[a], hi = (double)mid[a] + (double)half[a];
      if (hi < 0.0 || lo > top) { in = false; break; }
      b.lo[a] = (float)lo; if ((double)b.lo[a] > lo) b.lo[a] = std::nextafter(b.lo[a], -INFINITY);
      b.hi[a] = (float)hi; if ((double)b.hi[a] < hi) b.hi[a] = std::nextafter(b.hi[a], INFINITY);
    }
    if (!in) continue;
    used[(size_t)j] = 1;
    boxes[(size_t)j] = b;
    for (int a = 0; a < 2; ++a) {
      if (std::isfinite(b.lo[a])) h0[a] = std::min(h0[a], (double)b.lo[a]);
      if (std::isfinite(b.hi[a])) h1[a] = std::max(h1[a], (double)b.hi[a]);
    }
  }
  for (int a = 0; a < 2; ++a) {
    h0[a] = std::max(h0[a], 0.0); h1[a] = std::min(h1[a], 1.0);
    if (!(h1[a] - h0[a] > 1e-6)) { h0[a] = 0.0; h1[a] = 1.0; }
  }
  const size_t n_groups = ((size_t)n + 3) / 4;
  if (masks.size() < n_groups) masks.resize(n_groups);
  double best_cost = INFINITY;
  std::vector<uint32_t> trial(masks.size());
  for (uint32_t ns = 32; ns >= 1; ns /= 2) {
    const uint32_t nt = 32 / ns;
    TilePlan t;
    t.ns = ns; t.smax = (float)(ns - 1); t.tmax = (float)(nt - 1);
    t.ks = (float)(ns / (h1[0] - h0[0])); t.cs = (float)(-h0[0] * (double)t.ks);
    t.kt = (float)(nt / (h1[1] - h0[1])); t.ct = (float)(-h0[1] * (double)t.kt);
    std::fill(trial.begin(), trial.end(), 0u);
    for (int j = 0; j < n; ++j) {
      if (!used[(size_t)j]) continue;
      const Box& b = boxes[(size_t)j];
      const uint32_t c0 = tile_index(b.lo[0], t.ks, t.cs, t.smax), c1 = tile_index(b.hi[0], t.ks, t.cs, t.smax);
      const uint32_t r0 = tile_index(b.lo[1], t.kt, t.ct, t.tmax), r1 = tile_index(b.hi[1], t.kt, t.ct, t.tmax);
      uint32_t m = 0;
      for (uint32_t r = r0; r <= r1; ++r)
        for (uint32_t cc = c0; cc <= c1; ++cc) m |= 1u << (cc + ns * r);
      trial[(size_t)j / 4] |= m;
    }
    // share of the image each column / row covers (the edge ones take what lies outside the grid)
    double wide[32], high[32];
    for (uint32_t i = 0; i < ns; ++i) {
      const double a = i == 0 ? 0.0 : h0[0] + i * (h1[0] - h0[0]) / ns, b = i == ns - 1 ? 1.0 : h0[0] + (i + 1) * (h1[0] - h0[0]) / ns;
      wide[i] = std::max(0.0, b - a);
    }
    for (uint32_t i = 0; i < nt; ++i) {
      const double a = i == 0 ? 0.0 : h0[1] + i * (h1[1] - h0[1]) / nt, b = i == nt - 1 ? 1.0 : h0[1] + (i + 1) * (h1[1] - h0[1]) / nt;
      high[i] = std::max(0.0, b - a);
    }
    double cost = 0.0;
    for (size_t g = 0; g < n_groups; ++g)
      for (uint32_t m = trial[g]; m != 0; m &= m - 1) {
        const uint32_t bit = (uint32_t)__builtin_ctz(m);
        cost += wide[bit % ns] * high[bit / ns];
      }
    if (cost < best_cost) { best_cost = cost; plan = t; masks = trial; }
  }
}

static int refresh_spans(rtow_ctx* c) {
  c->spans_valid = false;
  if (!c->have_scene || !c->have_camera) return RTOW_OK;
  const SpanCamera sc = span_camera(c->cam);
  if (!sc.covered) return RTOW_OK;

  const int n = c->n_spheres, n_alloc = c->n_geom_pad + 4 * rtow::kSpanPadGroups;   // padding: see scan_interval
  // per list entry the middle, per group of four the largest half width; the ranges of s, then those of t
  c->host_span_mid.assign((size_t)n_alloc * 2, kSpanNever);
  c->host_span_half.assign((size_t)(n_alloc / 4) * 2, -1.0f);
  std::vector<float> own_half((size_t)n * 2, -1.0f);   // per list entry: its own half widths (s, then t), for the masks
  for (int axis = 0; axis < 2; ++axis) {
    float* const mids = c->host_span_mid.data() + (size_t)axis * (size_t)n_alloc;
    float* const halves = c->host_span_half.data() + (size_t)axis * (size_t)(n_alloc / 4);
    for (int j = 0; j < n; ++j) {
      if (!c->host_regular[(size_t)j]) continue;   // the |radius| >= kBigRadius spheres are tested by every ray anyway
      float mid, half;
      span_of_sphere(c->cam, sc, axis, c->host_geom64[(size_t)j], mid, half);
      mids[(size_t)j] = mid;
      own_half[(size_t)axis * (size_t)n + (size_t)j] = half;
      if (half > halves[(size_t)(j / 4)]) halves[(size_t)(j / 4)] = half;
    }
  }
  c->host_tile_mask.assign(((size_t)(c->n_geom_pad / 4) + 15) / 16 * 16 + rtow::kTileMaskPad, 0u);
  build_tile_masks(n, c->host_span_mid.data(), own_half.data(), c->host_span_mid.data() + n_alloc, own_half.data() + n,
                   c->tiles, c->host_tile_mask);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));   // a render may still be reading the old tables
  HIP_TRY(hipMemcpyAsync(c->span_mid.p, c->host_span_mid.data(), sizeof(float) * c->host_span_mid.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->span_half.p, c->host_span_half.data(), sizeof(float) * c->host_span_half.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->tile_mask.p, c->host_tile_mask.data(), sizeof(uint32_t) * c->host_tile_mask.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->spans_valid = true;
  return RTOW_OK;
}

extern "C" {

int rtow_interval_plan(const rtow_camera_f64* cam, int n, const double* cx, const double* cy, const double* cz,
                       const double* radius, float* mid_s, float* half_s, float* mid_t, float* half_t, int* covered) {
  if (!cam || n < 0 || (n > 0 && (!cx || !cy || !cz || !radius)))
    return fail(RTOW_ERR_INVALID, "rtow_interval_plan: NULL argument or n = %d", n);
  rtow::CameraParams64 c64;
  for (int k = 0; k < 3; ++k) {
    c64.origin[k] = cam->origin[k]; c64.llc[k] = cam->lower_left[k]; c64.hor[k] = cam->horizontal[k];
    c64.ver[k] = cam->vertical[k]; c64.u[k] = cam->u[k]; c64.v[k] = cam->v[k];
  }
  c64.lens_radius = cam->lens_radius;
  const SpanCamera sc = span_camera(c64);
  if (covered) *covered = sc.covered ? 1 : 0;
  if (!sc.covered) return RTOW_OK;
  for (int j = 0; j < n; ++j)
    for (int axis = 0; axis < 2; ++axis) {
      float mid = kSpanNever, half = -1.0f;
      if (std::fabs((float)radius[j]) < rtow::kBigRadius) span_of_sphere(c64, sc, axis, make_double4(cx[j], cy[j], cz[j], radius[j]), mid, half);
      float* const m = axis == 0 ? mid_s : mid_t;
      float* const h = axis == 0 ? half_s : half_t;
      if (m) m[j] = mid;
      if (h) h[j] = half;
    }
  return RTOW_OK;
}

int rtow_mask_plan(const rtow_camera_f64* cam, int n, const double* cx, const double* cy, const double* cz,
                   const double* radius, uint32_t* group_mask, float* grid, int* columns, int* covered) {
  if (!cam || n < 0 || (n > 0 && (!cx || !cy || !cz || !radius)))
    return fail(RTOW_ERR_INVALID, "rtow_mask_plan: NULL argument or n = %d", n);
  try {
    std::vector<float> spans((size_t)n * 4, 0.0f);
    float* const ms = spans.data(); float* const hs = ms + n; float* const mt = hs + n; float* const ht = mt + n;
    int cov = 0;
    const int rc = rtow_interval_plan(cam, n, cx, cy, cz, radius, ms, hs, mt, ht, &cov);
    if (rc != RTOW_OK) return rc;
    if (covered) *covered = cov;
    if (!cov) return RTOW_OK;
    TilePlan plan;
    std::vector<uint32_t> masks(((size_t)n + 3) / 4, 0u);
    build_tile_masks(n, ms, hs, mt, ht, plan, masks);
    if (group_mask) std::copy(masks.begin(), masks.begin() + ((size_t)n + 3) / 4, group_mask);
    if (grid) { grid[0] = plan.ks; grid[1] = plan.cs; grid[2] = plan.kt; grid[3] = plan.ct; grid[4] = plan.smax; grid[5] = plan.tmax; }
    if (columns) *columns = (int)plan.ns;
  } catch (const std::exception& e) {
    return fail(RTOW_ERR_NOMEM, "rtow_mask_plan: %s", e.what());
  }
  return RTOW_OK;
}

}  // extern "C"

template <typename T>
static int set_scene_impl(rtow_ctx* c, int n, const T* cx, const T* cy, const T* cz, const T* radius, const uint8_t* kind,
                          const T* albedo, const T* fuzz, const T* ir);

// the tables are staged in std::vector: an allocation failure must come back as a status, not unwind
// through the C boundary
template <typename T>
static int set_scene_guarded(rtow_ctx* c, int n, const T* cx, const T* cy, const T* cz, const T* radius,
                             const uint8_t* kind, const T* albedo, const T* fuzz, const T* ir) {
  try {
    return set_scene_impl<T>(c, n, cx, cy, cz, radius, kind, albedo, fuzz, ir);
  } catch (const std::bad_alloc&) {
    return fail(RTOW_ERR_NOMEM, "rtow_set_scene: out of host memory staging %d spheres", n);
  } catch (...) {
    return fail(RTOW_ERR_INVALID, "rtow_set_scene: unexpected exception");
  }
}

extern "C" {

int rtow_set_scene(rtow_ctx* c, int n, const float* cx, const float* cy, const float* cz,
                   const float* radius, const uint8_t* kind, const float* albedo, const float* fuzz,
                   const float* ir) {
  return set_scene_guarded<float>(c, n, cx, cy, cz, radius, kind, albedo, fuzz, ir);
}

int rtow_set_scene_f64(rtow_ctx* c, int n, const double* cx, const double* cy, const double* cz,
                       const double* radius, const uint8_t* kind, const double* albedo, const double* fuzz,
                       const double* ir) {
  return set_scene_guarded<double>(c, n, cx, cy, cz, radius, kind, albedo, fuzz, ir);
}

int rtow_set_hittables(rtow_ctx* c, int n, const uint8_t* hittable_kind, const double* cx, const double* cy,
                       const double* cz, const double* radius, const uint8_t* material_kind, const double* albedo,
                       const double* fuzz, const double* ir) {
  // hit*(h: Hittable) (hittables.nim:75-77) is a case over HittableKind with the one branch htSphere
  if (hittable_kind)
    for (int j = 0; j < n; ++j)
      if (hittable_kind[j] != RTOW_HT_SPHERE)
        return fail(RTOW_ERR_INVALID, "rtow_set_hittables: entry %d has HittableKind %d; the only kind is htSphere (%d)", j,
                    (int)hittable_kind[j], (int)RTOW_HT_SPHERE);
  return set_scene_guarded<double>(c, n, cx, cy, cz, radius, material_kind, albedo, fuzz, ir);
}

}  // extern "C"

template <typename T>
static int set_scene_impl(rtow_ctx* c, int n, const T* cx, const T* cy, const T* cz, const T* radius, const uint8_t* kind,
                          const T* albedo, const T* fuzz, const T* ir) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_set_scene: ctx is NULL");
  if (n < 0) return fail(RTOW_ERR_INVALID, "rtow_set_scene: n = %d", n);
  if (n > rtow::kMaxSpheres) return fail(RTOW_ERR_INVALID, "rtow_set_scene: n = %d (at most %d spheres)", n, rtow::kMaxSpheres);
  if (n > 0 && (!cx || !cy || !cz || !radius || !kind || !albedo || !fuzz || !ir))
    return fail(RTOW_ERR_INVALID, "rtow_set_scene: NULL array with n = %d", n);
  for (int j = 0; j < n; ++j) {
    if (kind[j] > RTOW_DIELECTRIC) return fail(RTOW_ERR_INVALID, "rtow_set_scene: kind[%d] = %d", j, (int)kind[j]);
    const float rf = (float)radius[j];   // the binary32 rounding must be usable too
    if (!(std::fabs(rf) > 0.0f) || !std::isfinite(rf))
      return fail(RTOW_ERR_INVALID, "rtow_set_scene: radius[%d] = %g", j, (double)radius[j]);
    if (!std::isfinite((float)cx[j]) || !std::isfinite((float)cy[j]) || !std::isfinite((float)cz[j]))
      return fail(RTOW_ERR_INVALID, "rtow_set_scene: center[%d] is not finite", j);
    // Radiance is summed in fixed point with a per-sample ceiling of 3.5 per channel.  With every albedo in
    // [0, 1] a sample never exceeds the sky's 1.0, so the ceiling is never met; an albedo above 1 (which the
    // reference would accept) could exceed it and the image would silently differ - refused instead.
    if (kind[j] != RTOW_DIELECTRIC)
      for (int k = 0; k < 3; ++k)
        if (!((float)albedo[3 * j + k] >= 0.0f && (float)albedo[3 * j + k] <= 1.0f))
          return fail(RTOW_ERR_INVALID, "rtow_set_scene: albedo[%d][%d] = %g is outside [0, 1]", j, k, (double)albedo[3 * j + k]);
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));  // a render may still be reading the old tables

  const int n_pad = ((n + rtow::kGeomPad - 1) / rtow::kGeomPad) * rtow::kGeomPad;
  // four spheres per 16-float group: cx0 cx1 cy0 cy1 cz0 cz1 | cx2 cx3 cy2 cy3 cz2 cz3 | k0 k1 k2 k3
  // with k = |c|^2 - r^2 (+inf: never passes the filter)
  // plus two all-padding groups behind the last one (the scan prefetches up to three groups ahead)
  const int n_alloc = n_pad + 8;
  std::vector<float> geom((size_t)n_alloc * 4, 0.0f);
  auto geom_at = [&geom](int j, int field) -> float& {
    if (field == 3) return geom[(size_t)(j / 4) * 16 + 12 + (size_t)(j % 4)];
    return geom[(size_t)(j / 4) * 16 + (size_t)((j % 4) / 2) * 6 + (size_t)field * 2 + (size_t)(j % 2)];
  };
  std::vector<rtow::SphereShade> shade((size_t)(n > 0 ? n : 1));
  std::vector<rtow::BigSphere> big;
  for (int j = 0; j < n_alloc; ++j) geom_at(j, 3) = INFINITY;
  std::vector<float4> exact((size_t)n_alloc, make_float4(0.f, 0.f, 0.f, 0.f));
  // (-k is also streamed through LDS in tiles of 256 entries by tables that do not fit there: whole tiles, and one more)
  std::vector<float> nk(((size_t)n_alloc + 255) / 256 * 256 + 256, -INFINITY);
  // the scene as given, for the binary64 tests of primary rays; every other table holds binary32 roundings
  std::vector<double4> geom64((size_t)(n > 0 ? n : 1), make_double4(0.0, 0.0, 0.0, 1.0));
  float cmax = 0.0f;
  // The slab test of the camera rays (rtow_kernels.hip, pair_slab) compares a ray's distance to each of a group's
  // four centres with ONE radius per group: the largest |r| among the group's spheres (a group without a sphere of the
  // binary32 table gets -1 and never passes).  Tables that fit LDS take -k from there, which leaves the last four
  // floats of a group record free: the last of them carries that radius; larger tables keep it in an array of its own.
  const bool slab_table = n_pad <= rtow::kMaxSpheresKInLds;
  const int n_groups_alloc = n_alloc / 4;
  std::vector<float> group_radius((size_t)n_groups_alloc, -1.0f);
  for (int j = 0; j < n; ++j) {
    geom64[j] = make_double4((double)cx[j], (double)cy[j], (double)cz[j], (double)radius[j]);
    const float fx = (float)cx[j], fy = (float)cy[j], fz = (float)cz[j];
    const float r = (float)radius[j];
    const bool is_big = std::fabs(r) >= rtow::kBigRadius;
    geom_at(j, 0) = fx; geom_at(j, 1) = fy; geom_at(j, 2) = fz;
    if (!is_big) {
      const double c2 = (double)fx * fx + (double)fy * fy + (double)fz * fz;
      geom_at(j, 3) = (float)(c2 - (double)r * r);
      exact[j] = make_float4(fx, fy, fz, r * r);
      nk[j] = -geom_at(j, 3);
      if (std::fabs(r) > group_radius[(size_t)(j / 4)]) group_radius[(size_t)(j / 4)] = std::fabs(r);
      const float reach = (float)std::sqrt(c2) + std::fabs(r);
      if (reach > cmax) cmax = reach;
    }
    if (is_big) {
      rtow::BigSphere b;
      b.cx = fx; b.cy = fy; b.cz = fz; b.r = r; b.rr = (double)r * (double)r; b.index = j; b.pad_ = 0;
      big.push_back(b);
    }
    rtow::SphereShade& s = shade[j];
    s.cx = fx; s.cy = fy; s.cz = fz; s.inv_r = 1.0f / r;
    s.ar = (float)albedo[3 * j]; s.ag = (float)albedo[3 * j + 1]; s.ab = (float)albedo[3 * j + 2]; s.fuzz = (float)fuzz[j];
    s.kind = kind[j]; s.ir = (float)ir[j]; s.inv_ir = 1.0f / s.ir; s.pad_ = 0.f;
    if (kind[j] == RTOW_DIELECTRIC) {
      // a dielectric has no albedo (hitrecord.nim:109): its slots carry Schlick's r0^2 (hitrecord.nim:103-104)
      // for the two refraction ratios a hit can see, in the binary32 operations the CPU mirror uses per hit
      s.ar = rtow::schlick_r0(s.inv_ir);   // front face: ratio = 1 / ir
      s.ag = rtow::schlick_r0(s.ir);       // back face:  ratio = ir
      s.ab = 0.f;
    }
  }
  // The test's slack is absolute (rounding of a value of the size of the scene) and enters as a scale of the ray,
  // 1 / (1 + slack / smallest group radius): radii below 2^-8 of the scene's extent are raised to that, so that one
  // tiny sphere cannot blunt the test for all the others (a larger radius only lets more groups through).
  float slab_rmin = INFINITY;
  for (int g = 0; g < n_groups_alloc; ++g) {
    float& rg = group_radius[(size_t)g];
    if (rg > 0.0f) {
      rg = std::max(rg, cmax * 0x1p-8f);
      slab_rmin = std::min(slab_rmin, rg);
    }
    if (slab_table)
      for (int k = 0; k < 4; ++k) geom[(size_t)g * 16 + 12 + (size_t)k] = k == 3 ? rg : 0.0f;
  }
  const std::vector<float>& slab = group_radius;
  HIP_TRY(c->geom64.reserve(sizeof(double4) * geom64.size()));
  HIP_TRY(hipMemcpyAsync(c->geom64.p, geom64.data(), sizeof(double4) * geom64.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c->geom.reserve(sizeof(float) * geom.size()));
  HIP_TRY(c->geom_rr.reserve(sizeof(float4) * exact.size()));
  HIP_TRY(c->geom_nk.reserve(sizeof(float) * nk.size()));
  HIP_TRY(c->span_mid.reserve(sizeof(float) * (size_t)(n_pad + 4 * rtow::kSpanPadGroups) * 2));        // filled by refresh_spans once a camera is known
  HIP_TRY(c->span_half.reserve(sizeof(float) * (size_t)(n_pad / 4 + rtow::kSpanPadGroups) * 2));
  HIP_TRY(c->tile_mask.reserve(sizeof(uint32_t) * (((size_t)(n_pad / 4) + 15) / 16 * 16 + rtow::kTileMaskPad)));
  HIP_TRY(c->geom_slab.reserve(sizeof(float) * slab.size()));
  HIP_TRY(hipMemcpyAsync(c->geom_slab.p, slab.data(), sizeof(float) * slab.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c->shade.reserve(sizeof(rtow::SphereShade) * shade.size()));
  HIP_TRY(c->big.reserve(sizeof(rtow::BigSphere) * (big.empty() ? 1 : big.size())));
  HIP_TRY(hipMemcpyAsync(c->geom.p, geom.data(), sizeof(float) * geom.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->geom_rr.p, exact.data(), sizeof(float4) * exact.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->geom_nk.p, nk.data(), sizeof(float) * nk.size(), hipMemcpyHostToDevice, c->stream));
  c->cmax = cmax;
  c->slab_inv_rmin = std::isfinite(slab_rmin) ? 1.0f / slab_rmin : 1.0f;
  HIP_TRY(hipMemcpyAsync(c->shade.p, shade.data(), sizeof(rtow::SphereShade) * shade.size(), hipMemcpyHostToDevice, c->stream));
  if (!big.empty())
    HIP_TRY(hipMemcpyAsync(c->big.p, big.data(), sizeof(rtow::BigSphere) * big.size(), hipMemcpyHostToDevice, c->stream));
  // The uploads are ordered on the context's stream, like the renders that read the tables; the staging vectors
  // above must outlive the copies.
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->n_spheres = n; c->n_geom_pad = n_pad; c->n_big = (int)big.size();
  c->blocks_per_cu = rtow::trace_blocks_per_cu(n_pad);
  if (const char* cap = getenv("RTOW_BLOCKS_PER_CU")) {  // experiments only: cap the resident workgroups per CU
    const int v = atoi(cap);
    if (v >= 1 && v < c->blocks_per_cu) c->blocks_per_cu = v;
  }
  c->have_scene = true;
  c->samples_done = 0;              // accumulated samples belong to the previous scene: no pass may continue them
  c->last_row_list_valid = false;
  c->host_geom64.assign(geom64.begin(), geom64.begin() + n);
  c->host_regular.assign((size_t)n, 0);
  for (int j = 0; j < n; ++j) c->host_regular[(size_t)j] = std::fabs((float)radius[j]) < rtow::kBigRadius ? 1 : 0;
  return refresh_spans(c);
}

extern "C" {

static int set_camera_f64(rtow_ctx* c, const rtow_camera_f64& cam) {
  rtow::CameraParams64 next;
  for (int k = 0; k < 3; ++k) {
    next.origin[k] = cam.origin[k]; next.llc[k] = cam.lower_left[k]; next.hor[k] = cam.horizontal[k];
    next.ver[k] = cam.vertical[k]; next.u[k] = cam.u[k]; next.v[k] = cam.v[k];
  }
  next.lens_radius = cam.lens_radius;
  // the same camera again (a host that sets it before every frame): the interval test's tables still match
  const bool tables_match = c->have_camera && c->spans_valid && std::memcmp(&next, &c->cam, sizeof(next)) == 0;
  c->cam = next;
  {
    // direction of the ray through the middle of the image; the slab test projects along the axis it has least of
    double d[3], best = INFINITY;
    for (int k = 0; k < 3; ++k) d[k] = cam.lower_left[k] + 0.5 * cam.horizontal[k] + 0.5 * cam.vertical[k] - cam.origin[k];
    c->slab_axis = 1;
    for (int k = 0; k < 3; ++k)
      if (std::fabs(d[k]) < best) { best = std::fabs(d[k]); c->slab_axis = k; }
  }
  c->have_camera = true;
  c->samples_done = 0;              // accumulated samples belong to the previous camera: no pass may continue them
  c->last_row_list_valid = false;
  if (tables_match) return RTOW_OK;
  try {   // refresh_spans stages its tables in host vectors: no exception may leave through the C boundary
    return refresh_spans(c);
  } catch (const std::bad_alloc&) {
    c->spans_valid = false;         // the camera itself is set; its rays see the slab test
    return fail(RTOW_ERR_NOMEM, "rtow_set_camera: out of host memory for the interval test's tables");
  } catch (...) {
    c->spans_valid = false;
    return fail(RTOW_ERR_INVALID, "rtow_set_camera: unexpected exception");
  }
}

int rtow_set_camera_f64(rtow_ctx* c, const rtow_camera_f64* cam) {
  if (!c || !cam) return fail(RTOW_ERR_INVALID, "rtow_set_camera_f64: NULL argument");
  return set_camera_f64(c, *cam);
}

int rtow_set_camera(rtow_ctx* c, const rtow_camera* cam) {
  if (!c || !cam) return fail(RTOW_ERR_INVALID, "rtow_set_camera: NULL argument");
  rtow_camera_f64 w;   // binary32 widens exactly
  for (int k = 0; k < 3; ++k) {
    w.origin[k] = cam->origin[k]; w.lower_left[k] = cam->lower_left[k]; w.horizontal[k] = cam->horizontal[k];
    w.vertical[k] = cam->vertical[k]; w.u[k] = cam->u[k]; w.v[k] = cam->v[k]; w.w[k] = cam->w[k];
  }
  w.lens_radius = cam->lens_radius;
  return set_camera_f64(c, w);
}

// where one frame's kernels read their row list and keep their sums, counters and pixels
struct FrameTargets {
  const int32_t* rows_dev = nullptr;      // device copy of rows[], or nullptr for rows 0..n_rows-1
  unsigned long long* accum = nullptr;    // n_rows*w*3 sums
  unsigned char* counters = nullptr;      // kCounterBytes: segment count, next path, clock stamps
  int32_t* hit_id = nullptr;              // n_rows*w, or nullptr
  uint8_t* rgb = nullptr;                 // n_rows*w*3
};

// clear + trace + resolve + counter read-back on the context's stream; allocates nothing
static int launch_frame(rtow_ctx* c, int w, int h, int spp, int max_depth, uint64_t seed, int first_sample, int n_rows,
                        const FrameTargets& t) {
  const uint64_t n_pixels = (uint64_t)n_rows * (uint64_t)w;
  const uint64_t total_paths = n_pixels * (uint64_t)spp;
  rtow::TraceParams p;
  p.cam = c->cam;
  p.geom = (const float*)c->geom.p; p.geom_exact = (const float4*)c->geom_rr.p; p.geom_nk = (const float*)c->geom_nk.p; p.n_geom_pad = c->n_geom_pad;
  p.geom64 = (const double4*)c->geom64.p; p.n_list = c->n_spheres;
  p.geom_slab = (const float*)c->geom_slab.p;
  p.span_mid = (const float*)c->span_mid.p; p.span_half = (const float*)c->span_half.p;
  p.tile_mask = (const uint32_t*)c->tile_mask.p;
  p.tile_ks = c->tiles.ks; p.tile_cs = c->tiles.cs; p.tile_kt = c->tiles.kt; p.tile_ct = c->tiles.ct;
  p.tile_smax = c->tiles.smax; p.tile_tmax = c->tiles.tmax; p.tile_ns = c->tiles.ns;
  p.cmax = c->cmax;
  p.slab_inv_rmin = c->slab_inv_rmin;
  p.pad_ = 0.0f;
  p.big = (const rtow::BigSphere*)c->big.p; p.n_big = c->n_big;
  p.shade = (const rtow::SphereShade*)c->shade.p;
  p.W = w; p.H = h; p.spp = spp; p.max_depth = max_depth; p.sample_base = (uint32_t)first_sample;
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  p.rows = t.rows_dev; p.n_rows = n_rows;
  // Measured (scripts/sweep_chunks.py): chunks of 192-256 paths beat 512 shrinking to 64 - by 0.5 % on the
  // full frame and 4 % on a 1/8 share, where the end of the frame weighs more - and 128 or less pay for
  // the chunk-to-pixel arithmetic (about 250 instructions per chunk).
  p.chunk_max = 256;
  p.chunk_min = 256; p.total_paths = total_paths;
  if (const char* e = getenv("RTOW_CHUNK_MAX")) {  // experiments only: multiples of 64
    const int v = atoi(e);
    if (v >= 64 && v <= 65536 && v % 64 == 0) p.chunk_max = (uint32_t)v;
  }
  if (const char* e = getenv("RTOW_CHUNK_MIN")) {
    const int v = atoi(e);
    if (v >= 64 && (uint32_t)v <= p.chunk_max && v % 64 == 0) p.chunk_min = (uint32_t)v;
  }
  p.inv_spp = (uint32_t)((0x100000000ull + (uint64_t)spp - 1) / (uint64_t)spp);
  p.seg_counter = (unsigned long long*)t.counters;
  p.path_counter = (unsigned long long*)(t.counters + 8);
  p.accum = t.accum;
  p.hit_id = t.hit_id;

  const uint64_t waves_needed = (total_paths + 63) / 64;
  int grid = c->n_cu * c->blocks_per_cu;
  const uint64_t blocks_needed = (waves_needed + (rtow::kTraceThreads / 64) - 1) / (rtow::kTraceThreads / 64);
  if ((uint64_t)grid > blocks_needed) grid = (int)(blocks_needed ? blocks_needed : 1);
  p.n_waves = (uint32_t)grid * (rtow::kTraceThreads / 64);
  uint32_t taper = 4;   // chunks shrink once fewer than this many chunks of chunk_max per wave are left
  if (const char* e = getenv("RTOW_CHUNK_TAPER")) {  // experiments only
    const int v = atoi(e);
    if (v >= 1 && v <= 4096) taper = (uint32_t)v;
  }
  while (taper > 1 && (uint64_t)taper * p.n_waves * p.chunk_max >= 0x100000000ull) taper /= 2;
  p.chunk_taper = taper * p.n_waves;

  HIP_TRY(hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(hipMemsetAsync(t.counters, 0, kCounterBytes, c->stream));
  if (first_sample == 0 && n_pixels) HIP_TRY(hipMemsetAsync(t.accum, 0, sizeof(unsigned long long) * 3 * n_pixels, c->stream));
  HIP_TRY(hipEventRecord(c->ev[1], c->stream));
  // What the camera rays of a wave see before the filter (rtow_kernels.hip): by default the mask test followed by the
  // interval test (scan_masked: the tile of the image the ray aims at against the tiles in which a group's spheres can
  // be met, then scan_interval's test - the ray's image coordinates against the ranges in which the sphere can be met -
  // on the groups that pass; tables from refresh_spans); the interval test alone for frames under kTileFrameMin pixels
  // wide or high; or, for a camera that construction does not cover, the slab test along the axis the view direction
  // is most perpendicular to (pair_slab).  In all of them every ray still tests every group of the list itself.  The
  // image does not depend on the choice; for measurements and tests: RTOW_NO_SLAB_TEST = nothing in front of the filter,
  // RTOW_CAMERA_SCREEN=interval = the interval test without the masks, RTOW_CAMERA_SCREEN=slab = the slab test,
  // RTOW_SLAB_AXIS=0|1|2 = the slab test along a given axis,
  // RTOW_CAMERA_SCREEN=bundle = the bundle test (scan_bundle: the table against the wave's 64 rays as a whole).  The
  // bundle test decides about a sphere once per wave instead of once per ray, which is culling by ray packets rather
  // than brute force: it is an option, never the default.
  const bool masks_apply = w >= rtow::kTileFrameMin && h >= rtow::kTileFrameMin;   // what build_tile_masks assumed of s and t
  int slab_axis = c->spans_valid ? (masks_apply ? rtow::kMaskTest : rtow::kIntervalTest) : c->slab_axis;
  if (getenv("RTOW_NO_SLAB_TEST")) slab_axis = -1;
  else if (const char* e = getenv("RTOW_SLAB_AXIS")) { const int v = atoi(e); if (v >= 0 && v <= 2) slab_axis = v; }
  else if (const char* s = getenv("RTOW_CAMERA_SCREEN")) {
    if (!strcmp(s, "bundle")) slab_axis = rtow::kBundleTest;
    else if (!strcmp(s, "slab")) slab_axis = c->slab_axis;
    else if (!strcmp(s, "interval") && c->spans_valid) slab_axis = rtow::kIntervalTest;
  }
  c->screen_launched = slab_axis < 0 ? RTOW_SCREEN_NONE : slab_axis == rtow::kMaskTest ? RTOW_SCREEN_MASK
                     : slab_axis == rtow::kIntervalTest ? RTOW_SCREEN_INTERVAL
                     : slab_axis == rtow::kBundleTest ? RTOW_SCREEN_BUNDLE : RTOW_SCREEN_SLAB;
  if (n_pixels) rtow::launch_trace(p, slab_axis, grid, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev[2], c->stream));
  rtow::launch_resolve(t.accum, t.rgb, (uint32_t)(3 * n_pixels), first_sample + spp, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev[3], c->stream));
  HIP_TRY(hipMemcpyAsync(c->host_counters, t.counters, kCounterBytes, hipMemcpyDeviceToHost, c->stream));
  c->pending = true;
  c->ppm_bytes = 0;
  c->last_w = w; c->last_h = h; c->last_spp = spp; c->last_rows = n_rows;
  c->last_seed = seed;
  return RTOW_OK;
}

// first_sample > 0: a further pass over the same pixels - the accumulators are kept, the launch traces
// samples first_sample .. first_sample + spp - 1 and the image is resolved over all of them
static int enqueue_render(rtow_ctx* c, int w, int h, int spp, int max_depth, uint64_t seed,
                          const int32_t* rows, int n_rows, void* rgb_dev, bool want_hit_id, int first_sample = 0) {
  if (!c) return fail(RTOW_ERR_INVALID, "render: ctx is NULL");
  if (!c->have_scene || !c->have_camera)
    return fail(RTOW_ERR_STATE, "render: call rtow_set_scene and rtow_set_camera first");
  if (w < 2 || h < 2) return fail(RTOW_ERR_INVALID, "render: image %dx%d (width and height must be >= 2)", w, h);
  if (spp < 1 || max_depth < 1) return fail(RTOW_ERR_INVALID, "render: spp = %d, max_depth = %d", spp, max_depth);
  if (n_rows < 0 || n_rows > h) return fail(RTOW_ERR_INVALID, "render: n_rows = %d of %d", n_rows, h);
  if (!rows && n_rows != h) return fail(RTOW_ERR_INVALID, "render: rows == NULL needs n_rows == h");
  if (rows)
    for (int r = 0; r < n_rows; ++r)
      if (rows[r] < 0 || rows[r] >= h) return fail(RTOW_ERR_INVALID, "render: rows[%d] = %d of %d", r, rows[r], h);
  if (first_sample < 0 || (int64_t)first_sample + spp > rtow::kMaxSamplesPerPixel)
    return fail(RTOW_ERR_INVALID, "render: samples %d .. %lld (a pixel's 64-bit sums hold at most %d samples)", first_sample,
                (long long)first_sample + spp - 1, rtow::kMaxSamplesPerPixel);
  if (first_sample > 0) {
    const bool same_rows = rows ? (c->last_row_list_valid && c->last_row_list.size() == (size_t)n_rows &&
                                   std::equal(c->last_row_list.begin(), c->last_row_list.end(), rows))
                                : (c->last_row_list_valid && c->last_row_list.empty());
    if (c->last_w != w || c->last_h != h || c->last_rows != n_rows || !same_rows || c->samples_done != first_sample ||
        c->last_seed != seed)
      return fail(RTOW_ERR_STATE, "render pass: samples %d.. must continue a render of the same image, rows and seed "
                                  "that stopped at sample %d", first_sample, c->samples_done);
  }
  const uint64_t n_pixels = (uint64_t)n_rows * (uint64_t)w;
  if (n_pixels >= 0x7fffffffull / 3) return fail(RTOW_ERR_INVALID, "render: %llu pixels in one call", (unsigned long long)n_pixels);

  HIP_TRY(hipSetDevice(c->device));
  if (c->pending) { HIP_TRY(hipStreamSynchronize(c->stream)); release_scratch(c); }

  try {
    if (first_sample == 0) {
      std::vector<int32_t> list;
      if (rows) list.assign(rows, rows + n_rows);
      c->last_row_list.swap(list);      // nothing below this line throws: the image state stays consistent
      c->last_row_list_valid = true;
    }
  } catch (const std::bad_alloc&) {
    return fail(RTOW_ERR_NOMEM, "render: out of host memory for %d row indices", n_rows);
  }
  const int rc0 = [&]() -> int {
    HIP_TRY(c->accum.reserve(sizeof(unsigned long long) * 3 * (n_pixels ? n_pixels : 1)));
    HIP_TRY(c->rgb.reserve(3 * (n_pixels ? n_pixels : 1)));
    if (want_hit_id) HIP_TRY(c->hit_id.reserve(sizeof(int32_t) * (n_pixels ? n_pixels : 1)));
    if (rows) {
      // uploaded from the context's own copy: the caller may release rows[] as soon as this call returns
      HIP_TRY(c->rows.reserve(sizeof(int32_t) * (size_t)(n_rows ? n_rows : 1)));
      HIP_TRY(hipMemcpyAsync(c->rows.p, c->last_row_list.data(), sizeof(int32_t) * (size_t)n_rows, hipMemcpyHostToDevice,
                             c->stream));
    }
    return RTOW_OK;
  }();
  if (rc0 != RTOW_OK) { c->samples_done = 0; c->last_row_list_valid = false; return rc0; }

  FrameTargets t;
  t.rows_dev = rows ? (const int32_t*)c->rows.p : nullptr;
  t.accum = (unsigned long long*)c->accum.p;
  t.counters = (unsigned char*)c->counters.p;
  t.hit_id = want_hit_id ? (int32_t*)c->hit_id.p : nullptr;
  t.rgb = rgb_dev ? (uint8_t*)rgb_dev : (uint8_t*)c->rgb.p;
  const int rc = launch_frame(c, w, h, spp, max_depth, seed, first_sample, n_rows, t);
  if (rc != RTOW_OK) { c->samples_done = 0; c->last_row_list_valid = false; return rc; }
  c->image_in_workspace = rgb_dev == nullptr;
  c->sums_in_workspace = true;
  c->samples_done = first_sample + spp;
  return RTOW_OK;
}

int rtow_sync(rtow_ctx* c) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_sync: ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  release_scratch(c);
  if (c->pending) {
    c->pending = false;
    c->stats.ray_segments = c->host_counters[0];
    c->stats.paths = (uint64_t)c->last_rows * (uint64_t)c->last_w * (uint64_t)c->last_spp;
    c->stats.n_spheres = c->n_spheres;
    c->stats.n_big = c->n_big;
    c->stats.gather_path = RTOW_GATHER_NONE;   // rtow_render_multi sets it once its rows are home
    c->stats.camera_screen = c->screen_launched;
    c->stats.reserved_ = 0;
    {
      const uint64_t ticks = c->host_counters[4] - c->host_counters[2], ref = c->host_counters[5] - c->host_counters[3];
      c->stats.shader_clock_mhz = ref > 0 && c->host_counters[5] > c->host_counters[3] ? (float)((double)ticks / (double)ref * 100.0) : 0.0f;
    }
    HIP_TRY(hipEventElapsedTime(&c->stats.trace_ms, c->ev[1], c->ev[2]));
    HIP_TRY(hipEventElapsedTime(&c->stats.total_ms, c->ev[0], c->ev[3]));
  }
  return RTOW_OK;
}

int rtow_render_async(rtow_ctx* c, int w, int h, int spp, int max_depth, uint64_t seed,
                      const int32_t* rows, int n_rows, void* rgb_dev) {
  return enqueue_render(c, w, h, spp, max_depth, seed, rows, n_rows, rgb_dev, false);
}

int rtow_render_pass(rtow_ctx* c, int w, int h, int first_sample, int n_samples, int max_depth, uint64_t seed,
                     const int32_t* rows, int n_rows, void* rgb_dev) {
  return enqueue_render(c, w, h, n_samples, max_depth, seed, rows, n_rows, rgb_dev, false, first_sample);
}

int rtow_render_noalloc(rtow_ctx* c, int w, int h, int spp, int max_depth, uint64_t seed, const int32_t* rows_dev,
                        int n_rows, void* rgb_dev) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_render_noalloc: ctx is NULL");
  if (!c->have_scene || !c->have_camera)
    return fail(RTOW_ERR_STATE, "render: call rtow_set_scene and rtow_set_camera first");
  if (!c->scratch_dev) return fail(RTOW_ERR_STATE, "rtow_render_noalloc: no work memory on device %d", c->device);
  if (!rgb_dev) return fail(RTOW_ERR_INVALID, "rtow_render_noalloc: rgb_dev is NULL (the image has nowhere to go)");
  if (w < 2 || h < 2) return fail(RTOW_ERR_INVALID, "render: image %dx%d (width and height must be >= 2)", w, h);
  if (spp < 1 || max_depth < 1) return fail(RTOW_ERR_INVALID, "render: spp = %d, max_depth = %d", spp, max_depth);
  if (spp > rtow::kMaxSamplesPerPixel)
    return fail(RTOW_ERR_INVALID, "render: spp = %d (a pixel's 64-bit sums hold at most %d samples)", spp, rtow::kMaxSamplesPerPixel);
  if (n_rows < 0 || n_rows > h) return fail(RTOW_ERR_INVALID, "render: n_rows = %d of %d", n_rows, h);
  if (!rows_dev && n_rows != h) return fail(RTOW_ERR_INVALID, "render: rows_dev == NULL needs n_rows == h");
  const uint64_t n_pixels = (uint64_t)n_rows * (uint64_t)w;
  if (n_pixels > kNoAllocMaxPixels)
    return fail(RTOW_ERR_INVALID, "rtow_render_noalloc: %llu pixels in one call, the fixed work memory holds %u "
                                  "(use rtow_render_async, or render the frame in row bands)",
                (unsigned long long)n_pixels, kNoAllocMaxPixels);
  if (c->device >= kMaxDevices) return fail(RTOW_ERR_STATE, "rtow_render_noalloc: device %d", c->device);
  HIP_TRY(hipSetDevice(c->device));
  if (c->pending) HIP_TRY(hipStreamSynchronize(c->stream));   // this context's events and counters are still in use
  {
    // one context at a time owns this GPU's work memory; it keeps it until it has been synchronised
    rtow_ctx* owner = nullptr;
    if (!g_scratch_owner[c->device].compare_exchange_strong(owner, c, std::memory_order_acq_rel) && owner != c)
      return fail(RTOW_ERR_STATE, "rtow_render_noalloc: the work memory of device %d is in use by a frame of another "
                                  "context; rtow_sync that context first", c->device);
    c->holds_scratch = true;
  }
  FrameTargets t;
  t.rows_dev = rows_dev;
  t.counters = c->scratch_dev;
  t.accum = (unsigned long long*)(c->scratch_dev + kScratchHeader);
  t.hit_id = nullptr;
  t.rgb = (uint8_t*)rgb_dev;
  // the sums live in shared work memory: no later pass may continue this image, and nothing stays in the context
  c->samples_done = 0;
  c->last_row_list_valid = false;
  c->image_in_workspace = false;
  c->sums_in_workspace = false;
  const int rc = launch_frame(c, w, h, spp, max_depth, seed, 0, n_rows, t);
  if (rc != RTOW_OK) { (void)hipStreamSynchronize(c->stream); release_scratch(c); }
  return rc;
}

int rtow_read_image(rtow_ctx* c, uint8_t* rgb_out, uint64_t* accum_out) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_read_image: ctx is NULL");
  if (c->last_w == 0) return fail(RTOW_ERR_STATE, "rtow_read_image: nothing has been rendered on this context");
  if (rgb_out && !c->image_in_workspace)
    return fail(RTOW_ERR_STATE, "rtow_read_image: the last render wrote its image to the caller's device buffer");
  if (accum_out && !c->sums_in_workspace)
    return fail(RTOW_ERR_STATE, "rtow_read_image: the last render kept its sums outside the context (rtow_render_noalloc)");
  const int rc = rtow_sync(c);
  if (rc != RTOW_OK) return rc;
  const size_t n_pixels = (size_t)c->last_rows * (size_t)c->last_w;
  if (rgb_out && n_pixels) { const int r2 = copy_to_host(c, rgb_out, c->rgb.p, 3 * n_pixels); if (r2 != RTOW_OK) return r2; }
  if (accum_out && n_pixels) return copy_to_host(c, accum_out, c->accum.p, sizeof(uint64_t) * 3 * n_pixels);
  return RTOW_OK;
}

int rtow_render(rtow_ctx* c, int w, int h, int spp, int max_depth, uint64_t seed, const int32_t* rows,
                int n_rows, uint8_t* rgb_out, uint64_t* accum_out, int32_t* hit_id_out) {
  int rc = enqueue_render(c, w, h, spp, max_depth, seed, rows, n_rows, nullptr, hit_id_out != nullptr);
  if (rc != RTOW_OK) return rc;
  rc = rtow_sync(c);
  if (rc != RTOW_OK) return rc;
  const size_t n_pixels = (size_t)n_rows * (size_t)w;
  if (rgb_out && n_pixels) { rc = copy_to_host(c, rgb_out, c->rgb.p, 3 * n_pixels); if (rc != RTOW_OK) return rc; }
  if (accum_out && n_pixels) {
    rc = copy_to_host(c, accum_out, c->accum.p, sizeof(uint64_t) * 3 * n_pixels);
    if (rc != RTOW_OK) return rc;
  }
  if (hit_id_out && n_pixels) return copy_to_host(c, hit_id_out, c->hit_id.p, sizeof(int32_t) * n_pixels);
  return RTOW_OK;
}

static int render_multi_impl(rtow_ctx* const* ctxs, int n_ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                             int tile_rows, uint8_t* rgb_out, uint64_t* ray_segments_out);

int rtow_render_multi(rtow_ctx* const* ctxs, int n_ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                      int tile_rows, uint8_t* rgb_out, uint64_t* ray_segments_out) {
  try {
    return render_multi_impl(ctxs, n_ctx, w, h, spp, max_depth, seed, tile_rows, rgb_out, ray_segments_out);
  } catch (const std::bad_alloc&) {
    if (ctxs) for (int k = 0; k < n_ctx; ++k) if (ctxs[k]) (void)rtow_sync(ctxs[k]);
    return fail(RTOW_ERR_NOMEM, "rtow_render_multi: out of host memory");
  } catch (...) {
    return fail(RTOW_ERR_INVALID, "rtow_render_multi: unexpected exception");
  }
}

}  // extern "C"

// The dealing of rows behind rtow_render_multi (also exported as rtow_multi_plan): row y belongs to context
// (y / tile_rows) mod n_ctx - sky rows cost one segment per path and ground rows several, so contiguous bands
// would be unbalanced - and a context's rows keep their image order.
struct MultiPlan {
  std::vector<std::vector<int32_t>> rows;   // per context: its image rows, ascending
  size_t row_bytes = 0;                     // 3 * w
  size_t slot = 0;                          // bytes per context in the gather buffer: the largest share, rounded up to 256
};

static MultiPlan multi_plan(int w, int h, int n_ctx, int tile_rows) {
  MultiPlan p;
  p.rows.resize((size_t)n_ctx);
  for (int y = 0; y < h; ++y) p.rows[(size_t)((y / tile_rows) % n_ctx)].push_back(y);
  p.row_bytes = 3 * (size_t)w;
  for (const auto& r : p.rows) p.slot = std::max(p.slot, p.row_bytes * r.size());
  p.slot = (p.slot + 255) & ~(size_t)255;
  return p;
}

// The RCCL clique of an ordered device list: looked up, or created on first use and kept for the life of the
// process.  Called BEFORE a frame is enqueued - communicator construction launches kernels of its own and must
// not meet GPUs that a persistent trace grid occupies.  nullptr (with the reason in `why`) when RCCL cannot be used.
// The caller holds g_rccl_mutex.
static std::vector<ncclComm_t>* rccl_clique(const std::vector<int>& devs, std::string& why) {
  if (const char* missing = rccl_load()) { why = missing; return nullptr; }
  std::vector<ncclComm_t>& comms = g_rccl.comms[devs];
  if (comms.empty()) {
    comms.resize(devs.size());
    const ncclResult_t e = g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data());
    if (e != ncclSuccess) {
      why = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e);
      g_rccl.comms.erase(devs);
      return nullptr;
    }
  }
  return &comms;
}

// The framebuffer gather of rtow_render_multi over RCCL.  Every context k > 0 sends its rows (they are still being
// rendered: the send is ordered behind the render on that context's stream) to the first context's GPU, which
// receives them side by side; all sends and receives are enqueued in one group before anything is waited for.  Then
// one copy into pinned host memory and the rows are dealt into the caller's image.  Every context is drained before
// this returns, whatever happens; a failure leaves rgb_out untouched, so the caller can still copy the rows - which
// are complete in every context's own image buffer - through the host.
static int gather_rccl(rtow_ctx* const* ctxs, int n_ctx, const MultiPlan& plan, std::vector<ncclComm_t>& comms,
                       uint8_t* rgb_out, uint64_t* ray_segments_out) {
  const size_t row_bytes = plan.row_bytes, slot = plan.slot;
  rtow_ctx* c0 = ctxs[0];
  int rc = RTOW_OK;
  auto body = [&]() -> int {
    // tests only: the exchange "fails" before anything is sent, so that the way back through the host copies runs
    if (getenv("RTOW_RCCL_INJECT_FAILURE")) return fail(RTOW_ERR_DEVICE, "rtow_render_multi: RCCL failure injected by the environment");
    HIP_TRY(hipSetDevice(c0->device));
    HIP_TRY(c0->gather.reserve(slot * (size_t)n_ctx ? slot * (size_t)n_ctx : 256));
    if (c0->gather_host_cap < slot * (size_t)n_ctx) {
      if (c0->gather_host) { (void)hipHostFree(c0->gather_host); c0->gather_host = nullptr; c0->gather_host_cap = 0; }
      HIP_TRY(hipHostMalloc((void**)&c0->gather_host, slot * (size_t)n_ctx, hipHostMallocDefault));
      c0->gather_host_cap = slot * (size_t)n_ctx;
    }
    ncclResult_t e = g_rccl.GroupStart();
    for (int k = 1; k < n_ctx && e == ncclSuccess; ++k) {
      const size_t bytes = row_bytes * plan.rows[(size_t)k].size();
      if (bytes == 0) continue;
      e = g_rccl.Send(ctxs[k]->rgb.p, bytes, ncclUint8, 0, comms[(size_t)k], ctxs[k]->stream);
      if (e == ncclSuccess) e = g_rccl.Recv((char*)c0->gather.p + slot * (size_t)k, bytes, ncclUint8, k, comms[0], c0->stream);
    }
    const ncclResult_t e2 = g_rccl.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return fail(RTOW_ERR_DEVICE, "rtow_render_multi: RCCL send/receive: %s", g_rccl.GetErrorString(e));
    // the first context's own rows join the others on its own stream, behind its render and the receives
    HIP_TRY(hipSetDevice(c0->device));
    const size_t own = row_bytes * plan.rows[0].size();
    if (own) HIP_TRY(hipMemcpyAsync(c0->gather.p, c0->rgb.p, own, hipMemcpyDeviceToDevice, c0->stream));
    HIP_TRY(hipMemcpyAsync(c0->gather_host, c0->gather.p, slot * (size_t)n_ctx, hipMemcpyDeviceToHost, c0->stream));
    return RTOW_OK;
  };
  rc = body();
  uint64_t segments = 0;
  for (int k = 0; k < n_ctx; ++k) {   // every context is drained even after a failure
    const int r2 = rtow_sync(ctxs[k]);
    if (r2 != RTOW_OK && rc == RTOW_OK) rc = r2;
    segments += ctxs[k]->stats.ray_segments;
  }
  if (rc != RTOW_OK) return rc;
  for (int k = 0; k < n_ctx; ++k) {
    const std::vector<int32_t>& r = plan.rows[(size_t)k];
    for (size_t i = 0; i < r.size(); ++i)
      memcpy(rgb_out + row_bytes * (size_t)r[i], c0->gather_host + slot * (size_t)k + row_bytes * i, row_bytes);
  }
  if (ray_segments_out) *ray_segments_out = segments;
  return RTOW_OK;
}

static int render_multi_impl(rtow_ctx* const* ctxs, int n_ctx, int w, int h, int spp, int max_depth, uint64_t seed,
                             int tile_rows, uint8_t* rgb_out, uint64_t* ray_segments_out) {
  if (!ctxs || n_ctx < 1) return fail(RTOW_ERR_INVALID, "rtow_render_multi: no contexts");
  if (tile_rows < 1) return fail(RTOW_ERR_INVALID, "rtow_render_multi: tile_rows = %d", tile_rows);
  if (w < 2 || h < 2) return fail(RTOW_ERR_INVALID, "render: image %dx%d (width and height must be >= 2)", w, h);
  for (int k = 0; k < n_ctx; ++k) {
    if (!ctxs[k]) return fail(RTOW_ERR_INVALID, "rtow_render_multi: ctxs[%d] is NULL", k);
    for (int j = 0; j < k; ++j)
      if (ctxs[j] == ctxs[k]) return fail(RTOW_ERR_INVALID, "rtow_render_multi: ctxs[%d] and ctxs[%d] are the same context", j, k);
  }
  const MultiPlan plan = multi_plan(w, h, n_ctx, tile_rows);
  // How the rows come home.  Default: every context's rows are copied to the host in turn.  RTOW_MULTI_GATHER=rccl asks for
  // the device-to-device gather over RCCL (xGMI) onto the first context's GPU; it needs pairwise different GPUs (a clique
  // of one context is allowed: nothing is sent).  The communicators are looked up or built HERE, before any context
  // has work enqueued.  Whatever goes wrong on that way, the host copies still deliver the frame.
  bool all_different = true;
  for (int k = 0; k < n_ctx; ++k)
    for (int j = 0; j < k; ++j) all_different = all_different && ctxs[j]->device != ctxs[k]->device;
  const char* mode = getenv("RTOW_MULTI_GATHER");
  const bool want_rccl = mode && !strcmp(mode, "rccl") && all_different && rgb_out;
  int path = RTOW_GATHER_HOST;
  std::unique_lock<std::mutex> rccl_lock(g_rccl_mutex, std::defer_lock);
  std::vector<ncclComm_t>* comms = nullptr;
  if (want_rccl) {
    rccl_lock.lock();   // collectives on one clique are serialised; held until the gather has been waited for
    std::vector<int> devs((size_t)n_ctx);
    for (int k = 0; k < n_ctx; ++k) devs[(size_t)k] = ctxs[k]->device;
    std::string why;
    comms = rccl_clique(devs, why);
    if (!comms) { rccl_lock.unlock(); path = RTOW_GATHER_HOST_AFTER_RCCL; }
  }
  for (int k = 0; k < n_ctx; ++k) {
    static const int32_t no_rows[1] = {0};  // a context left without rows still takes part (and reports zero segments)
    const std::vector<int32_t>& r = plan.rows[(size_t)k];
    const int rc = enqueue_render(ctxs[k], w, h, spp, max_depth, seed, r.empty() ? no_rows : r.data(), (int)r.size(),
                                  nullptr, false);
    if (rc != RTOW_OK) {
      for (int j = 0; j < k; ++j) (void)rtow_sync(ctxs[j]);
      return rc;
    }
  }
  if (comms) {
    const int rc = gather_rccl(ctxs, n_ctx, plan, *comms, rgb_out, ray_segments_out);
    rccl_lock.unlock();
    if (rc == RTOW_OK) {
      for (int k = 0; k < n_ctx; ++k) ctxs[k]->stats.gather_path = RTOW_GATHER_RCCL;
      return RTOW_OK;
    }
    path = RTOW_GATHER_HOST_AFTER_RCCL;   // every context has been drained; its rows are intact in its own image buffer
  }
  uint64_t segments = 0;
  int first_error = RTOW_OK;
  std::vector<uint8_t> staging;
  const size_t row_bytes = plan.row_bytes;
  for (int k = 0; k < n_ctx; ++k) {
    rtow_ctx* c = ctxs[k];
    int rc = rtow_sync(c);   // every context is drained even after a failure
    const std::vector<int32_t>& r = plan.rows[(size_t)k];
    if (rc == RTOW_OK && rgb_out && !r.empty()) {
      staging.resize(row_bytes * r.size());
      const hipError_t e = hipSetDevice(c->device);
      if (e != hipSuccess) rc = fail(RTOW_ERR_DEVICE, "rtow_render_multi: %s", hipGetErrorString(e));
      else rc = copy_to_host(c, staging.data(), c->rgb.p, staging.size());
      if (rc == RTOW_OK)
        for (size_t i = 0; i < r.size(); ++i)
          memcpy(rgb_out + row_bytes * (size_t)r[i], staging.data() + row_bytes * i, row_bytes);
    }
    c->stats.gather_path = path;
    if (rc == RTOW_OK) segments += c->stats.ray_segments;
    else if (first_error == RTOW_OK) first_error = rc;
  }
  if (first_error != RTOW_OK) return first_error;
  if (ray_segments_out) *ray_segments_out = segments;
  return RTOW_OK;
}

extern "C" {

int rtow_multi_plan(int w, int h, int n_ctx, int tile_rows, int32_t* ctx_of_row, int32_t* index_in_ctx,
                    int32_t* rows_per_ctx, uint64_t* slot_bytes) {
  if (w < 1 || h < 0 || n_ctx < 1 || tile_rows < 1)
    return fail(RTOW_ERR_INVALID, "rtow_multi_plan: w = %d, h = %d, n_ctx = %d, tile_rows = %d", w, h, n_ctx, tile_rows);
  try {
    const MultiPlan p = multi_plan(w, h, n_ctx, tile_rows);
    for (int k = 0; k < n_ctx; ++k) {
      const std::vector<int32_t>& r = p.rows[(size_t)k];
      if (rows_per_ctx) rows_per_ctx[k] = (int32_t)r.size();
      for (size_t i = 0; i < r.size(); ++i) {
        if (ctx_of_row) ctx_of_row[r[i]] = k;
        if (index_in_ctx) index_in_ctx[r[i]] = (int32_t)i;
      }
    }
    if (slot_bytes) *slot_bytes = p.slot;
  } catch (const std::bad_alloc&) {
    return fail(RTOW_ERR_NOMEM, "rtow_multi_plan: out of host memory");
  }
  return RTOW_OK;
}

}  // extern "C"

extern "C" {

int rtow_format_ppm(rtow_ctx* c, int binary, char* out, uint64_t capacity, uint64_t* length) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_format_ppm: ctx is NULL");
  if (!out && !length) return fail(RTOW_ERR_INVALID, "rtow_format_ppm: out and length are both NULL");
  if (c->last_w == 0) return fail(RTOW_ERR_STATE, "rtow_format_ppm: nothing has been rendered on this context");
  if (!c->image_in_workspace)
    return fail(RTOW_ERR_STATE, "rtow_format_ppm: the last render wrote its image to the caller's device buffer");
  int rc = rtow_sync(c);
  if (rc != RTOW_OK) return rc;
  char header[64];
  const int header_len = snprintf(header, sizeof(header), "P%c\n%d %d\n255\n", binary ? '6' : '3', c->last_w, c->last_rows);
  const uint64_t n_pixels = (uint64_t)c->last_w * (uint64_t)c->last_rows;
  uint64_t body = 3 * n_pixels;
  if (!binary) {
    if (c->ppm_bytes == 0 && n_pixels) {
      const uint32_t n_blocks = (uint32_t)((n_pixels + rtow::kPpmBlock - 1) / rtow::kPpmBlock);
      HIP_TRY(c->ppm_len.reserve(sizeof(uint32_t) * n_blocks));
      HIP_TRY(c->ppm_off.reserve(sizeof(unsigned long long) * ((size_t)n_blocks + 1)));
      rtow::launch_ppm_lengths((const uint8_t*)c->rgb.p, (uint32_t)n_pixels, (uint32_t*)c->ppm_len.p, c->stream);
      rtow::launch_ppm_scan((const uint32_t*)c->ppm_len.p, n_blocks, (unsigned long long*)c->ppm_off.p, c->stream);
      HIP_TRY(hipGetLastError());
      unsigned long long total = 0;
      HIP_TRY(hipMemcpyAsync(&total, (const unsigned long long*)c->ppm_off.p + n_blocks, sizeof(total),
                             hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
      HIP_TRY(c->ppm_text.reserve((size_t)total));
      rtow::launch_ppm_text((const uint8_t*)c->rgb.p, (uint32_t)n_pixels, (const unsigned long long*)c->ppm_off.p,
                            (char*)c->ppm_text.p, c->stream);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipStreamSynchronize(c->stream));
      c->ppm_bytes = total;
    }
    body = c->ppm_bytes;
  }
  const uint64_t total_len = (uint64_t)header_len + body;
  if (length) *length = total_len;
  if (!out) return RTOW_OK;
  if (capacity < total_len)
    return fail(RTOW_ERR_INVALID, "rtow_format_ppm: %llu bytes needed, capacity %llu", (unsigned long long)total_len,
                (unsigned long long)capacity);
  memcpy(out, header, (size_t)header_len);
  if (body) return copy_to_host(c, out + header_len, binary ? c->rgb.p : c->ppm_text.p, (size_t)body);
  return RTOW_OK;
}

// shared body of rtow_trace_rays and rtow_get_rays: in_a / in_b are n*stride doubles each
static int run_rays(rtow_ctx* c, const char* who, bool get_rays, int n, const double* in_a, const double* in_b, int stride,
                    double t_min, double t_max, int max_depth, uint64_t seed, double* out3_a, int32_t* out_id,
                    double* out_b, int out_b_stride) {
  HIP_TRY(hipSetDevice(c->device));
  if (c->pending) { const int rc = rtow_sync(c); if (rc != RTOW_OK) return rc; }
  if (n == 0) return RTOW_OK;
  const size_t in_bytes = sizeof(double) * (size_t)n * (size_t)stride;
  const size_t a_bytes = sizeof(double) * 3 * (size_t)n, id_bytes = sizeof(int32_t) * (size_t)n;
  const size_t b_bytes = sizeof(double) * (size_t)out_b_stride * (size_t)n;
  HIP_TRY(c->ray_in.reserve(2 * in_bytes));
  HIP_TRY(c->ray_out.reserve(a_bytes + b_bytes + id_bytes));
  HIP_TRY(hipMemcpyAsync(c->ray_in.p, in_a, in_bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync((char*)c->ray_in.p + in_bytes, in_b, in_bytes, hipMemcpyHostToDevice, c->stream));
  rtow::RayParams p;
  p.cam = c->cam;
  p.orig = (const double*)c->ray_in.p; p.dir = (const double*)((char*)c->ray_in.p + in_bytes);
  p.n = n; p.t_min = t_min; p.t_max = t_max; p.max_depth = max_depth;
  p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  p.geom_nk = (const float*)c->geom_nk.p; p.geom_exact = (const float4*)c->geom_rr.p; p.geom64 = (const double4*)c->geom64.p;
  p.big = (const rtow::BigSphere*)c->big.p; p.n_big = c->n_big;
  p.shade = (const rtow::SphereShade*)c->shade.p; p.n_list = c->n_spheres; p.cmax = c->cmax;
  p.color = (get_rays || out3_a) ? (double*)c->ray_out.p : nullptr;
  p.t = (get_rays || out_b) ? (double*)((char*)c->ray_out.p + a_bytes) : nullptr;
  p.hit_id = out_id ? (int32_t*)((char*)c->ray_out.p + a_bytes + b_bytes) : nullptr;
  rtow::launch_rays(p, get_rays, c->stream);
  HIP_TRY(hipGetLastError());
  if (out3_a) HIP_TRY(hipMemcpyAsync(out3_a, p.color, a_bytes, hipMemcpyDeviceToHost, c->stream));
  if (out_b) HIP_TRY(hipMemcpyAsync(out_b, p.t, b_bytes, hipMemcpyDeviceToHost, c->stream));
  if (out_id) HIP_TRY(hipMemcpyAsync(out_id, p.hit_id, id_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  (void)who;
  return RTOW_OK;
}

int rtow_trace_rays(rtow_ctx* c, int n, const double* origins, const double* directions, double t_min, double t_max,
                    int max_depth, uint64_t seed, double* color_out, int32_t* hit_id_out, double* t_out) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_trace_rays: ctx is NULL");
  if (!c->have_scene) return fail(RTOW_ERR_STATE, "rtow_trace_rays: call rtow_set_scene first");
  if (n < 0 || (n > 0 && (!origins || !directions))) return fail(RTOW_ERR_INVALID, "rtow_trace_rays: n = %d or NULL rays", n);
  if (n > (1 << 27)) return fail(RTOW_ERR_INVALID, "rtow_trace_rays: n = %d rays in one call (at most 2^27)", n);
  if (!(t_min <= t_max)) return fail(RTOW_ERR_INVALID, "rtow_trace_rays: t_min = %g, t_max = %g", t_min, t_max);
  if (max_depth < 0) return fail(RTOW_ERR_INVALID, "rtow_trace_rays: max_depth = %d", max_depth);
  return run_rays(c, "rtow_trace_rays", false, n, origins, directions, 3, t_min, t_max, max_depth, seed, color_out, hit_id_out,
                  t_out, 1);
}

int rtow_get_rays(rtow_ctx* c, int n, const double* s, const double* t, uint64_t seed, double* origins_out,
                  double* directions_out) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_get_rays: ctx is NULL");
  if (!c->have_camera) return fail(RTOW_ERR_STATE, "rtow_get_rays: call rtow_set_camera first");
  if (n < 0 || (n > 0 && (!s || !t || !origins_out || !directions_out)))
    return fail(RTOW_ERR_INVALID, "rtow_get_rays: n = %d or NULL array", n);
  if (n > (1 << 27)) return fail(RTOW_ERR_INVALID, "rtow_get_rays: n = %d rays in one call (at most 2^27)", n);
  return run_rays(c, "rtow_get_rays", true, n, s, t, 1, 0.0, 0.0, 0, seed, origins_out, nullptr, directions_out, 3);
}

int rtow_scatter_rays(rtow_ctx* c, int n, const double* directions, const double* normals, const uint8_t* front_face,
                      const uint8_t* kind, const double* albedo_rgb, const double* fuzz, const double* ir, uint64_t seed,
                      double* attenuation_out, double* scattered_dir_out, uint8_t* scattered_out) {
  if (!c) return fail(RTOW_ERR_INVALID, "rtow_scatter_rays: ctx is NULL");
  if (n < 0 || (n > 0 && (!directions || !normals || !front_face || !kind || !albedo_rgb || !fuzz || !ir ||
                          !attenuation_out || !scattered_dir_out || !scattered_out)))
    return fail(RTOW_ERR_INVALID, "rtow_scatter_rays: n = %d or NULL array", n);
  if (n > (1 << 27)) return fail(RTOW_ERR_INVALID, "rtow_scatter_rays: n = %d triples in one call (at most 2^27)", n);
  for (int i = 0; i < n; ++i) {
    if (kind[i] > RTOW_DIELECTRIC) return fail(RTOW_ERR_INVALID, "rtow_scatter_rays: material %d has unknown kind %d", i, (int)kind[i]);
    if (kind[i] == RTOW_DIELECTRIC && !(std::isfinite(ir[i]) && ir[i] > 0.0))
      return fail(RTOW_ERR_INVALID, "rtow_scatter_rays: material %d: refraction index %g", i, ir[i]);
  }
  HIP_TRY(hipSetDevice(c->device));
  if (c->pending) { const int rc = rtow_sync(c); if (rc != RTOW_OK) return rc; }
  if (n == 0) return RTOW_OK;
  // device layout: [dir 3n | normal 3n | albedo 3n | fuzz n | ir n] doubles, then [front n | kind n] bytes
  const size_t N = (size_t)n, d3 = sizeof(double) * 3 * N, d1 = sizeof(double) * N;
  HIP_TRY(c->ray_in.reserve(3 * d3 + 2 * d1 + 2 * N));
  HIP_TRY(c->ray_out.reserve(2 * d3 + N));
  char* in = (char*)c->ray_in.p;
  char* out = (char*)c->ray_out.p;
  const void* src[7] = {directions, normals, albedo_rgb, fuzz, ir, front_face, kind};
  const size_t len[7] = {d3, d3, d3, d1, d1, N, N};
  size_t off[7], at = 0;
  for (int k = 0; k < 7; ++k) {
    off[k] = at;
    HIP_TRY(hipMemcpyAsync(in + at, src[k], len[k], hipMemcpyHostToDevice, c->stream));
    at += len[k];
  }
  rtow::ScatterParams p;
  p.n = n; p.seed_lo = (uint32_t)seed; p.seed_hi = (uint32_t)(seed >> 32);
  p.dir = (const double*)(in + off[0]); p.normal = (const double*)(in + off[1]); p.albedo = (const double*)(in + off[2]);
  p.fuzz = (const double*)(in + off[3]); p.ir = (const double*)(in + off[4]);
  p.front = (const uint8_t*)(in + off[5]); p.kind = (const uint8_t*)(in + off[6]);
  p.attenuation = (double*)out; p.out_dir = (double*)(out + d3); p.ok = (uint8_t*)(out + 2 * d3);
  rtow::launch_scatter(p, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(attenuation_out, p.attenuation, d3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(scattered_dir_out, p.out_dir, d3, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(scattered_out, p.ok, N, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOW_OK;
}

int rtow_selftest_arithmetic(rtow_ctx* c, uint64_t mismatches[2]) {
  if (!c || !mismatches) return fail(RTOW_ERR_INVALID, "rtow_selftest_arithmetic: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  if (c->pending) { const int rc = rtow_sync(c); if (rc != RTOW_OK) return rc; }
  HIP_TRY(c->counters.reserve(16));
  HIP_TRY(hipMemsetAsync(c->counters.p, 0, 16, c->stream));
  rtow::launch_arithmetic_selftest((unsigned long long*)c->counters.p, c->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(mismatches, c->counters.p, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RTOW_OK;
}

int rtow_get_stats(rtow_ctx* c, rtow_stats* out) {
  if (!c || !out) return fail(RTOW_ERR_INVALID, "rtow_get_stats: NULL argument");
  *out = c->stats;
  return RTOW_OK;
}

int rtow_get_stream(rtow_ctx* c, void** stream) {
  if (!c || !stream) return fail(RTOW_ERR_INVALID, "rtow_get_stream: NULL argument");
  *stream = (void*)c->stream;
  return RTOW_OK;
}

}  // extern "C"
