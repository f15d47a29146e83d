// gfx950 kernels of the renderer.
//
//  rtow_trace_kernel   persistent waves; every lane carries one path (pixel, sample) through an
//                      iterative bounce loop.  A wave alternates between iterations that start 64 new
//                      paths (camera ray and first hit in binary64, as the reference computes them) and
//                      iterations that trace the next binary32 segment of the paths in flight, idle
//                      lanes taking scattered paths from a per-wave pool in LDS first, so every sphere
//                      scan runs with full wavefronts.  Work is handed out in chunks (runs of
//                      consecutive paths, pixel-major) from a global counter.  The radiance of a path
//                      that ends is added in 64-bit fixed point (through a small per-wave cache in LDS)
//                      to the frame with integer atomics, which makes the image independent of
//                      scheduling, chunking and GPU count.
//  rtow_resolve_kernel mean + sqrt gamma + clamp + 8-bit quantisation (raytracer.nim:22-28).
//
// The sphere scan (hittables.nim:79-88 over hitrecord.nim:55-63) keeps the ray in VGPRs and
// streams the sphere table through SGPRs with wave-uniform scalar loads, four spheres per load,
// two spheres per packed FP32 instruction, and a rarely taken branch for discriminant >= 0; camera
// rays - 64 samples of one pixel per wave - see a cheaper per-ray test first (scan_interval; pair_slab for cameras
// the interval test does not cover).
//  rtow_rays_kernel, rtow_get_rays_kernel, rtow_scatter_kernel   the reference's per-ray procs (rayColor / hit,
//                      getRay, scatter) for batches of caller-given rays, one lane per ray.
//  ppm_*_kernel        the P3 text of the image, composed on the device.
#include <hip/hip_runtime.h>
#include <cstddef>
#include "rtow_kernels.h"

namespace rtow {

#ifdef RTOW_PROFILE_COUNTERS
// debug build only, summed over waves: [0] slots of the pixel-sum cache written out to the frame, [3] wave iterations, [4] lanes with a
// path over the BOUNCE iterations, [14] PRIMARY iterations, [9] groups with a survivor in some lane, [10] / [12]
// rounds of binary32 / binary64 exact tests, [11] passes of the shared rejection loop; clock cycles:
// [5] hand-out, [1] [2] [13] set-up, scan, shading of PRIMARY iterations, [6] [7] [8] of BOUNCE iterations, [15] all;
// lanes: [16] BOUNCE segments ending on the sky, [17] [18] [19] hitting a Lambertian / metal / dielectric sphere,
// [20] lanes still looking for a unit-ball point, summed over those passes, [21] [22] PRIMARY rays missing / hitting, [23] lanes taking a pooled path,
// [24] groups of a PRIMARY scan that go on to the filter (past the slab, bundle or interval tests), [25] groups that pass the
// interval test's first axis
__device__ unsigned long long g_prof[32];

// wave-local accumulation (values are wave-uniform), flushed once when the wave retires
#define RTOW_PROF_DECL unsigned long long prof_[32] = {0}
#define RTOW_PROF_PARAM , unsigned long long (&prof_)[32]
#define RTOW_PROF_ARG , prof_
#define RTOW_COUNT(i, v) do { prof_[i] += (unsigned long long)(v); } while (0)
#define RTOW_PROF_FLUSH() do { if ((threadIdx.x & 63u) == 0) for (int i_ = 0; i_ < 32; ++i_) atomicAdd(&g_prof[i_], prof_[i_]); } while (0)
#define RTOW_CLOCK() __builtin_amdgcn_s_memtime()
#elif defined(RTOW_PROFILE_TIMELINE)
// timeline build: per-wave time stamps only (plain stores to distinct addresses, no shared counters)
#define RTOW_PROF_DECL
#define RTOW_PROF_PARAM
#define RTOW_PROF_ARG
#define RTOW_COUNT(i, v) do { } while (0)
#define RTOW_PROF_FLUSH() do { } while (0)
#define RTOW_CLOCK() __builtin_amdgcn_s_memtime()
#else
#define RTOW_PROF_DECL
#define RTOW_PROF_PARAM
#define RTOW_PROF_ARG
#define RTOW_COUNT(i, v) do { } while (0)
#define RTOW_PROF_FLUSH() do { } while (0)
#define RTOW_CLOCK() 0ull
#endif

#ifdef RTOW_PROFILE_TIMELINE
// per wave, 12 slots: realtime stamps (100 MHz) at start [0], when the work counter ran dry [1], at
// the end [2]; live lanes [3] and iterations [4] at that point, iterations at the end [5]; core-clock
// stamps of the last iteration [6..9]; realtime at loop exit [10] and at the end of that iteration [11]
__device__ unsigned long long g_wave_times[12 * 16384];
#endif

#ifdef RTOW_GUARD
// debug build: every index into global memory is checked; a violation is recorded (site, value, limit)
// and the access skipped, so a wrong index shows up as a report instead of a GPU fault
__device__ unsigned long long g_guard[4 * 16];
static __device__ __forceinline__ bool guard_ok(int site, long long v, long long limit) {
  if (v >= 0 && v < limit) return true;
  if (atomicAdd(&g_guard[4 * site], 1ull) == 0ull) { g_guard[4 * site + 1] = (unsigned long long)v; g_guard[4 * site + 2] = (unsigned long long)limit; }
  return false;
}
#define RTOW_GUARDED(site, v, limit) guard_ok(site, (long long)(v), (long long)(limit))
#else
#define RTOW_GUARDED(site, v, limit) true
#endif

// A kernel argument read from the argument segment at the point of use (one scalar load) instead of being kept in a
// scalar register for the life of the wave.  The trace kernel has more wave-uniform state than scalar registers:
// what does not fit is parked in the lanes of a vector register and fetched back with v_readlane - VALU
// instructions in a kernel bound by VALU issue.  The empty asm makes the base address opaque, so the load cannot
// be hoisted out of the loop it sits in.
template <typename T>
static __device__ __forceinline__ T karg_load(size_t offset) {
  typedef const char __attribute__((address_space(4)))* base_ptr;
  typedef const T __attribute__((address_space(4)))* field_ptr;
  base_ptr base = (base_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(base));
  return *(field_ptr)(base + offset);
}
#define KARG(field) karg_load<decltype(TraceParams::field)>(offsetof(TraceParams, field))

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
// constant address space: wave-uniform loads from it become s_load_dwordx16 into SGPRs
typedef const f16v __attribute__((address_space(4)))* geom_ptr;
typedef float f4v __attribute__((ext_vector_type(4)));

// per-ray operands of the packed filter: seven scalars and the threshold in four register pairs.  A packed
// instruction can take either half of a source pair for both of its lanes (op_sel / op_sel_hi), so no
// operand has to be stored twice.
struct ScanRay {
  f2v uxy;     // unit direction x, y
  f2v uzn;     // unit direction z, -(o.u)
  f2v oxy;     // 2*o x, y
  f2v ozt;     // 2*o z, threshold = |o|^2 - margin
};

static __device__ __forceinline__ ScanRay scan_ray(f3 o, f3 u, float cmax) {
  ScanRay r;
  const FilterRay f = filter_ray(o, u, cmax);
  r.uxy = f2v{u.x, u.y}; r.uzn = f2v{u.z, f.nou};
  r.oxy = f2v{f.ox2, f.oy2}; r.ozt = f2v{f.oz2, f.threshold};
  return r;
}

// a * b + c on two spheres at once; a = a pair of sphere coordinates in scalar registers, b = ONE half of
// the pair `b` (B_HI selects which) used for both spheres, c = an accumulator pair, or one half of `c`
// for both (C_SEL: 0 = pair as it is, 1 = low half twice, 2 = high half twice)
template <int B_HI, int C_SEL>
static __device__ __forceinline__ f2v pk_fma_bcast(f2v a, f2v b, f2v c) {
  f2v d;
  if (B_HI == 0 && C_SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "s"(a), "v"(b), "v"(c));
  if (B_HI == 1 && C_SEL == 0) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(d) : "s"(a), "v"(b), "v"(c));
  if (B_HI == 0 && C_SEL == 2) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "s"(a), "v"(b), "v"(c));
  static_assert((B_HI == 0 || B_HI == 1) && (C_SEL == 0 || (C_SEL == 2 && B_HI == 0)), "forms the filter uses");
  return d;
}

// Slab test in front of the filter, for scans whose 64 rays are nearly the same ray (the camera rays of a
// PRIMARY iteration are consecutive samples of one pixel).  Component k of (c - o) x u is the distance between
// the sphere's centre and the ray's line as seen along axis k, times a factor <= 1: a sphere the ray hits has
//     | c_i u_j - c_j u_i - (o_i u_j - o_j u_i) | <= |r|          (i, j = the two other axes, in cyclic order).
// Two packed instructions per pair of spheres give the value for every (ray, sphere) pair, and the smallest |value|
// of a group's four spheres is compared with ONE radius per group, the largest |r| among the four (a scalar that
// comes with the group record): per group four packed FMAs, two minima and a compare - seven instructions against
// the filter's seventeen - and the group goes on to the filter only if some lane's ray passes, which for 64 samples
// of one pixel is one group in eleven on the book scene.  The host picks k per launch: the axis the view direction
// is most perpendicular to (the projection then shortens the rays least).
// Rounding.  The value evaluated in binary32 on rounded operands is off by less than 2^-21 (|o| + cmax); the test
// allows delta = 2^-14 (|o|_1 + cmax), 128 times that.  An absolute slack next to a per-group radius would cost an
// addition per group, so it enters as a scale of the RAY instead: the ray's operands (u_i, u_j, m) are multiplied by
// s = 1 / (1 + delta / rmin), rmin = the smallest group radius of the table, and |s value| <= rg is implied by
// |value| <= rg + delta for every rg >= rmin.  Every ray is still tested against every sphere; the test drops no
// sphere that is hit.
// sl = s (u_i, u_j), pre.x = s (o_j u_i - o_i u_j); returns s * value for the two spheres.
static __device__ __forceinline__ f2v pair_slab(const f2v ci, const f2v cj, const f2v sl, const f2v pre) {
  f2v t;
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(t) : "s"(cj), "v"(sl), "v"(pre));
  return pk_fma_bcast<1, 0>(ci, sl, t);
}

// min(|a.x|, |a.y|, |b.x|, |b.y|): the source modifiers take the absolute values
static __device__ __forceinline__ float min_abs4(f2v a, f2v b) {
  float m;
  asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(a.x), "v"(a.y), "v"(b.x));
  asm("v_min_f32_e64 %0, |%1|, %2" : "=v"(m) : "v"(b.y), "v"(m));
  return m;
}

// The ray's operands of pair_slab for axis AXIS: sl in the result's .lo, pre in its .hi.
struct SlabRay { f2v sl, pre; };
template <int AXIS>
static __device__ __forceinline__ SlabRay slab_operands(f3 o, f3 u, float cmax, float inv_rmin) {
  const float oi = AXIS == 0 ? o.y : AXIS == 1 ? o.z : o.x, oj = AXIS == 0 ? o.z : AXIS == 1 ? o.x : o.y;
  const float ui = AXIS == 0 ? u.y : AXIS == 1 ? u.z : u.x, uj = AXIS == 0 ? u.z : AXIS == 1 ? u.x : u.y;
  const float delta = (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + cmax) * 0x1p-14f;
  // (the hardware reciprocal is within 1 ulp; the factor below keeps s on the safe side of it)
  const float s = __builtin_amdgcn_rcpf(fmaf(delta, inv_rmin, 1.0f)) * (1.0f - 0x1p-20f);
  SlabRay r;
  r.sl = f2v{s * ui, s * uj};
  r.pre = f2v{s * fmaf(oj, ui, -(oi * uj)), 0.0f};
  return r;
}

static __device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Closest hit over the regular (binary32) table: a cheap packed filter over every sphere, and the
// exact Sphere.hit arithmetic (hitrecord.nim:56-63) only for the spheres the filter lets through.
//
// Filter.  With n = c.u - o.u and k = |c|^2 - r^2 the discriminant is
//     disc = n^2 - (k + |o|^2 - 2 o.c),
// so  disc >= -margin  <=>  n^2 + (2o).c - k >= |o|^2 - margin.   The left side costs eight
// packed instructions per PAIR of spheres (no per-sphere subtraction of the centre), the right
// side is one number per ray.  Rounding makes the filter value differ from the exact
// discriminant by a few ulps of (|o| + |c|)^2, which `margin` (the threshold in ScanRay::ozt) covers; a sphere
// that passes is then tested exactly, so the filter only ever adds work, never hits.
//
// The table is stored in groups of four spheres, 16 floats per group, as two packed pairs and k:
//   cx0 cx1  cy0 cy1  cz0 cz1 | cx2 cx3  cy2 cy3  cz2 cz3 | k0 k1 k2 k3
// Scalar loads bring a group into SGPRs, two groups ahead of its use: the first 12 floats when -k
// comes from LDS (scalar registers are the scarcest resource of this kernel), all 16 otherwise.  Spheres handled
// elsewhere and the padding carry k = +inf, so they never pass.  r^2 for the exact test comes from
// a separate array, fetched only when needed.
// K_SGPR: -k sits in scalar registers (tables too large for the LDS copy).  A VALU instruction reads at most one
// scalar operand, so -k cannot be the addend of an FMA whose multiplicand is a scalar centre coordinate; moved to
// vector registers first it costs two instructions per pair of spheres.  It is added LAST instead - the second
// chain starts as a product, n^2 + (2o).c is formed, and one packed add brings in -k: one instruction per pair.
// The value differs from the other form's by rounding only (the terms have the same magnitudes), far inside the
// margin; which spheres a screen lets through decides nothing but how many exact tests run.
template <bool K_SGPR = false>
static __device__ __forceinline__ f2v pair_filter(const f2v cx, const f2v cy, const f2v cz, const f2v nk,
                                                  const ScanRay& r) {
  f2v n = pk_fma_bcast<0, 2>(cx, r.uxy, r.uzn);   // cx * u.x - o.u
  n = pk_fma_bcast<1, 0>(cy, r.uxy, n);
  n = pk_fma_bcast<0, 0>(cz, r.uzn, n);
  if (K_SGPR) {
    f2v t, w;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "s"(cx), "v"(r.oxy));   // cx * 2o.x
    t = pk_fma_bcast<1, 0>(cy, r.oxy, t);
    t = pk_fma_bcast<0, 0>(cz, r.ozt, t);
    w = __builtin_elementwise_fma(n, n, t) + nk;   // left to the compiler (a packed add with nk as its scalar operand):
    return w;                                       // it then knows the sum is no signalling NaN and does not quiet it again
  }
  f2v t = pk_fma_bcast<0, 0>(cx, r.oxy, nk);      // nk = -k, a VGPR pair (see K_LDS)
  t = pk_fma_bcast<1, 0>(cy, r.oxy, t);
  t = pk_fma_bcast<0, 0>(cz, r.ozt, t);
  return __builtin_elementwise_fma(n, n, t);
}

// Spheres that pass the filter are not tested on the spot (only a few lanes of the wave would be
// busy).  Survivors go onto ONE list per wave in LDS as (lane, list index) words, and the list is
// worked off by all 64 lanes together - lane L takes entries L, L + 64, ... whoever they belong to:
// it fetches the owner's ray across the wave (ds_bpermute), runs the exact test and offers the root to
// the owner's closest-hit word, a sortable 64-bit key in LDS updated with ds_min_u64.  A wave has
// about 80 survivors per scan, unevenly spread over its lanes: per-lane queues took as many rounds as
// the longest of them (3.9 on average), the shared list takes total / 64 (1.5 on average).
// The closest-hit rule (smaller t, then larger index) does not depend on the order of the offers.
typedef unsigned long long __attribute__((address_space(3)))* lds_u64_ptr;
typedef uint32_t __attribute__((address_space(3)))* lds_u32_ptr;
typedef const double __attribute__((address_space(3)))* lds_f64_ptr;
// entries of a wave's survivor list; it is drained when fewer than 256 (one group, all lanes) are free.  Tables that do
// not fit LDS use 32-bit entries and fewer of them: their workgroups also hold the rings of scan_tiled, and five
// workgroups per CU must stay within 32 000 bytes of LDS each.
constexpr uint32_t kListEntriesShort = 512, kListEntriesLong = 352;

// An entry of the survivor list is (owner lane, list index).  Tables small enough to sit in LDS
// entirely (<= kMaxSpheresExactInLds entries) use 16-bit entries - the list then takes 1 KB per wave
// instead of 2 KB, which is what lets five workgroups per CU keep their tables in LDS.
template <bool SHORT> struct ListEntry;
template <> struct ListEntry<true> { typedef uint16_t type; static constexpr uint32_t kShift = 10; static constexpr uint32_t kEntries = kListEntriesShort; };
template <> struct ListEntry<false> { typedef uint32_t type; static constexpr uint32_t kShift = 26; static constexpr uint32_t kEntries = kListEntriesLong; };
static_assert(kMaxSpheresExactInLds + 16 <= (1 << ListEntry<true>::kShift), "list index must fit beside the lane number");
static_assert(kMaxSpheres + 16 <= (1 << ListEntry<false>::kShift), "list index must fit beside the lane number");

// per-wave LDS, in dwords: survivor list, 64 closest-hit keys (8 bytes), 64 closest-hit indices (primary
// segments), the path pool, the cache of pixel sums
template <bool SHORT> struct WaveLds {
  static constexpr uint32_t kList = SHORT ? kListEntriesShort / 2 : kListEntriesLong;
  static constexpr uint32_t kKeys = kList;            // 128 dwords
  static constexpr uint32_t kIds = kKeys + 128;       // 64 dwords
  static constexpr uint32_t kPool = kIds + 64;        // kPoolFields * 64 dwords, 8-byte aligned
  static constexpr uint32_t kSums = kPool + kPoolFields * 64;
  static constexpr uint32_t kTotal = kSums + 32;
  static_assert(kPool % 2 == 0 && kKeys % 2 == 0 && kSums % 2 == 0, "64-bit words must be aligned");
};

// What the exact tests of a PRIMARY scan need: the binary64 camera rays of the wave (LDS, six
// doubles per lane, field-major: ox oy oz dx dy dz) and the list index of every lane's closest hit so
// far (+1; 0 = none) beside its 64-bit key (the bits of t).  The binary64 sphere table is a kernel
// argument, read where it is used.
struct PrimaryCtx {
  lds_f64_ptr rays;
  lds_u32_ptr best_id;
};

template <bool SHORT>
struct CandList {
  typedef typename ListEntry<SHORT>::type entry_t;
  entry_t* list;      // LDS: ListEntry<SHORT>::kEntries entries of this wave
  lds_u64_ptr best;   // LDS: 64 words of this wave, lane L's closest hit so far (hit_key, or the bits of a binary64 t)
  uint32_t n;         // wave-uniform: entries in the list
  uint32_t lane;
};

// smaller t first, then the larger list index; id = -1 (nothing hit) loses every tie
// (unsigned arithmetic: id = -1 maps to 0x80000000, which signed subtraction may not produce)
static __device__ __forceinline__ unsigned long long hit_key(float t, int id) {
  return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(0x7fffffffu - (uint32_t)id);
}
static __device__ __forceinline__ int hit_key_id(unsigned long long key) { return (int)(0x7fffffffu - (uint32_t)key); }

static __device__ __forceinline__ void lds_order() {
  // one wave's LDS requests are served in order; the compiler only has to keep them in order too
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// randomInUnitSphere (basetypes.nim:114-118) for the lanes of a wave that need it (`need`), worked on by all 64.
// A lane's rejection loop is a search for the first accepted trip t = 0, 1, ... of its stream, trip t
// using draws 3t .. 3t+2 from the stream's position; any lane can evaluate any trip of any stream.  Left to
// themselves the lanes run as many passes as the unluckiest of them (5.3 with 26 owners, of which one in
// seven lanes does useful work); here the owners put their streams into LDS and the 64 lanes try T trips
// for each of them at once (T = the power of two that fits, 64 / owners), the owner takes the first
// accepted one from the lane that found it and advances its stream past exactly the trips the sequential
// loop would have used - so the result is the sequential loop's, bit for bit.
// Must be called by all 64 lanes together.  `slots`: 64 x 8 bytes of this wave's LDS, free at this point.
static __device__ __forceinline__ void unit_ball_wave(bool need, Stream& g, f3& b, float& l2, uint32_t lane, uint2* slots RTOW_PROF_PARAM) {
  constexpr uint32_t kTrip = 3u * kGolden;   // a trip is three draws
  b = {0.0f, 0.0f, 0.0f};
  l2 = 0.0f;
  uint64_t m = __ballot(need);
  while (m != 0) {
    const uint32_t n = (uint32_t)__popcll(m);
    RTOW_COUNT(11, 1);
    RTOW_COUNT(20, n);
    if (n > 32u) {
      // more owners than half the wave: each runs one trip of its own
      if (need) {
        f3 p;
        p.x = draw_pm1(g); p.y = draw_pm1(g); p.z = draw_pm1(g);
        const float lp = dot3(p, p);
        if (lp < 1.0f) { b = p; l2 = lp; need = false; }
      }
    } else {
      const uint32_t cl = 32u - (uint32_t)__clz((int)(n - 1u));   // ceil(log2 n)
      const uint32_t sh = cl == 0u ? 5u : 6u - cl;                // T = 32, 32, 16, 8, 4, 2 for n = 1, 2, <= 4, <= 8, <= 16, <= 32
      const uint32_t T = 1u << sh;
      const uint32_t r = lanes_below(m);                           // an owner's rank among the owners
      if (need) slots[r] = make_uint2(g.x0, g.key1);
      lds_order();
      const uint32_t q = lane >> sh;                               // the owner this lane works for
      const uint2 os = slots[q];
      Stream h = {os.x + (lane & (T - 1u)) * kTrip, os.y};
      f3 p;
      p.x = draw_pm1(h); p.y = draw_pm1(h); p.z = draw_pm1(h);
      const float lp = dot3(p, p);
      const uint64_t acc = __ballot(q < n && lp < 1.0f);
      const uint32_t mine = (uint32_t)(acc >> ((r << sh) & 63u)) & (0xffffffffu >> (32u - T));   // trips r*T .. r*T+T-1
      const bool found = need && mine != 0u;
      const uint32_t k = found ? (uint32_t)__builtin_ctz(mine) : 0u;
      const int src = (int)((((r << sh) + k) & 63u) << 2);
      const float fx = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, p.x)));
      const float fy = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, p.y)));
      const float fz = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, p.z)));
      const float fl = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, lp)));
      if (need) {
        g.x0 += (found ? k + 1u : T) * kTrip;
        if (found) { b = {fx, fy, fz}; l2 = fl; need = false; }
      }
      lds_order();
    }
    m = __ballot(need);
  }
}

// wave-uniform control flow; `pass` differs per lane
template <bool SHORT>
static __device__ __forceinline__ void cand_push(CandList<SHORT>& c, bool pass, int j) {
  // Usually one of a group's four spheres has survivors; the wave-uniform branch skips the other three
  // pushes (instructions issued with an empty execution mask still take their issue slots).
  const uint64_t m = __ballot(pass);
  if (m != 0) {
    if (pass) c.list[c.n + lanes_below(m)] = (typename CandList<SHORT>::entry_t)((c.lane << ListEntry<SHORT>::kShift) | (uint32_t)j);
    c.n += (uint32_t)__popcll(m);
  }
}

static __device__ __forceinline__ float from_lane(int byte_index, float v) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_index, __float_as_int(v)));
}

// Sphere.hit (hitrecord.nim:55-71) for a primary ray, operation for operation in binary64: a = |dir|^2
// is live, the direction is not normalised, t_min = 0.001 (raytracer.nim:11).  t_max is handled by
// the caller's minimum (hittables.nim:84-88 shrinks it as it goes, which selects the same root).
static __device__ __forceinline__ bool sphere_hit_f64(double ox, double oy, double oz, double dx, double dy, double dz,
                                                      double a, double4 sp, double& root) {
  const double ocx = ox - sp.x, ocy = oy - sp.y, ocz = oz - sp.z;
  double half_b = 0.0;   // dot accumulates from 0.0 in index order (basetypes.nim:25-27)
  half_b += ocx * dx; half_b += ocy * dy; half_b += ocz * dz;
  const double c = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w * sp.w;
  const double disc = half_b * half_b - a * c;
  if (disc < 0) return false;
  const double sqrtd = sqrt(disc);
  root = (-half_b - sqrtd) / a;
  if (root < 0.001) {
    root = (-half_b + sqrtd) / a;
    if (root < 0.001) return false;
  }
  return true;
}

// Must run with all 64 lanes active (rays are read from their owners' registers).
// PRIMARY: the list belongs to a batch of camera rays - the exact test is the reference's binary64
// one on the rays in pc.rays, and a lane's closest hit is a 64-bit t plus an index word (smaller t wins,
// equal t goes to the later list entry, hittables.nim:84-88).
template <bool PRIMARY, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void cand_drain(CandList<SHORT>& c, ExactPtr exact, f3 o, f3 u, float tmin,
                                                  int prev, const PrimaryCtx& pc RTOW_PROF_PARAM) {
  constexpr uint32_t kShift = ListEntry<SHORT>::kShift;
  lds_order();
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.n);   // a scalar loop bound
  for (uint32_t base = 0; base < n; base += 64u) {
    RTOW_COUNT(PRIMARY ? 12 : 10, 1);
    const uint32_t k = base + c.lane;
    const bool has = k < n;
    const uint32_t entry = c.list[k];   // lanes past the end read stale entries: unused
    const uint32_t owner = (entry >> kShift) & 63u;
    const int j = (int)(entry & ((1u << kShift) - 1u));
    if (PRIMARY) {
      bool valid = false;
      unsigned long long tb = ~0ull;
      if (has && RTOW_GUARDED(0, j, KARG(n_list))) {
        const lds_f64_ptr r = pc.rays + owner;
        const double dx = r[192], dy = r[256], dz = r[320];
        const double a = dx * dx + dy * dy + dz * dz;   // length_squared (basetypes.nim:88)
        double root;
        valid = sphere_hit_f64(r[0], r[64], r[128], dx, dy, dz, a, KARG(geom64)[j], root);
        tb = (unsigned long long)__double_as_longlong(root);
      }
      // t > 0, so the bit patterns order like the values.  A lane that lowers the key voids the index
      // word of the previous holder; every lane whose t is the key afterwards then offers its index.
      unsigned long long old = 0;
      if (valid) old = __hip_atomic_fetch_min(c.best + owner, tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      lds_order();
      const bool winner = valid && c.best[owner] == tb;
      if (winner && old > tb) pc.best_id[owner] = 0u;
      lds_order();
      if (winner) __hip_atomic_fetch_max(pc.best_id + owner, (uint32_t)j + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
      const int ob = (int)(owner << 2);
      const f3 oo = {from_lane(ob, o.x), from_lane(ob, o.y), from_lane(ob, o.z)};
      const f3 uu = {from_lane(ob, u.x), from_lane(ob, u.y), from_lane(ob, u.z)};
      const float tm = from_lane(ob, tmin);
      const int pv = __builtin_amdgcn_ds_bpermute(ob, prev);
      if (has) {
        const f4v sp = exact[j];   // (cx, cy, cz, r^2)
        const float ocx = oo.x - sp.x, ocy = oo.y - sp.y, ocz = oo.z - sp.z;
        float b = ocx * uu.x; b = fmaf(ocy, uu.y, b); b = fmaf(ocz, uu.z, b);
        float m = fmaf(-ocx, ocx, sp.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
        const float disc = fmaf(b, b, m);
        float t;
        const bool ok = regular_candidate(b, disc, tm, j == pv, t);
        if (disc >= 0.0f && ok)
          __hip_atomic_fetch_min(c.best + owner, hit_key(t, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
  }
  c.n = 0;
  lds_order();
}

struct GroupRec {   // centres of four spheres as packed pairs
  f8v a;            // cx0 cx1 cy0 cy1 cz0 cz1 cx2 cx3
  f4v b;            // cy2 cy3 cz2 cz3
  float rg;         // the group's radius for the slab test (loaded only for scans that run it)
};

typedef const float __attribute__((address_space(4)))* slab_ptr;   // the slab test's radius, one per group

template <bool K_LDS, bool SLAB = false>
static __device__ __forceinline__ GroupRec load_group(geom_ptr geom, int g, f4v& nk, slab_ptr slab = nullptr) {
  GroupRec r;
  if (K_LDS) {
    typedef const f8v __attribute__((address_space(4)))* p8;
    typedef const f4v __attribute__((address_space(4)))* p4;
    r.a = *(p8)(geom + g);
    r.b = *((p4)(geom + g) + 2);
    // tables whose -k sits in LDS keep the slab test's radius in the LAST float of the record: next to the centres
    // the compiler would fetch it with them as one eight-dword load, whose unused registers it then takes for
    // temporaries - and every write to one of those waits for the load
    if (SLAB) r.rg = *((slab_ptr)(geom + g) + 15);
  } else {
    const f16v v = geom[g];
    r.a = v.lo;
    r.b = f4v{v.s8, v.s9, v.sa, v.sb};
    nk = f4v{-v.sc, -v.sd, -v.se, -v.sf};
    if (SLAB) r.rg = slab[g];   // these tables use all 16 floats of a group: the radius comes from an array of its own
  }
  return r;
}

typedef const f4v __attribute__((address_space(3)))* nk_lds_ptr;

// nk: -k of the four spheres, or (NK_LATE) nk_late points at them in LDS and they are fetched only if the group
// gets as far as the filter
template <bool PRIMARY, int AXIS, bool NK_LATE, bool K_SGPR, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void group_test(const GroupRec grp, f4v nk, nk_lds_ptr nk_late, int j, ExactPtr exact,
                                                  CandList<SHORT>& cand, const ScanRay& r, const SlabRay& sr, f3 o, f3 u, float tmin,
                                                  int prev, const PrimaryCtx& pc RTOW_PROF_PARAM) {
  if (AXIS >= 0) {
    // coordinates i and j of the four centres, as the pairs the group holds them in
    const f2v ci0 = AXIS == 0 ? grp.a.s23 : AXIS == 1 ? grp.a.s45 : grp.a.s01, cj0 = AXIS == 0 ? grp.a.s45 : AXIS == 1 ? grp.a.s01 : grp.a.s23;
    const f2v ci1 = AXIS == 0 ? grp.b.xy : AXIS == 1 ? grp.b.zw : grp.a.s67, cj1 = AXIS == 0 ? grp.b.zw : AXIS == 1 ? grp.a.s67 : grp.b.xy;
    const f2v g0 = pair_slab(ci0, cj0, sr.sl, sr.pre);
    const f2v g1 = pair_slab(ci1, cj1, sr.sl, sr.pre);
    if (__ballot(min_abs4(g0, g1) <= grp.rg) == 0) return;   // wave-uniform: no ray of the wave comes near any of the four
    RTOW_COUNT(24, 1);
    if (NK_LATE) nk = *nk_late;
  }
  const f2v f0 = pair_filter(grp.a.s01, grp.a.s23, grp.a.s45, nk.xy, r);
  const f2v f1 = pair_filter(grp.a.s67, grp.b.xy, grp.b.zw, nk.zw, r);
  const float mx = fmaxf(fmaxf(f0.x, f0.y), fmaxf(f1.x, f1.y));
  if (__ballot(mx >= r.ozt.y) != 0) {   // wave-uniform: the list's fill count stays a scalar
    RTOW_COUNT(9, 1);
    cand_push(cand, f0.x >= r.ozt.y, j);
    cand_push(cand, f0.y >= r.ozt.y, j + 1);
    cand_push(cand, f1.x >= r.ozt.y, j + 2);
    cand_push(cand, f1.y >= r.ozt.y, j + 3);
    if (cand.n > ListEntry<SHORT>::kEntries - 256u) cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
  }
}

// One trip of a scan without the slab test: the two groups g and g + 1 (in ga / gb, their -k in ka / kb) are screened
// and the largest of the eight values is compared with the threshold - three v_max3 and a v_max instead of twice
// v_max + v_max3, one v_cmp instead of two: 33 instead of 34 instructions per eight spheres.  A group's scalar
// registers are free as soon as its fourteen FMAs are done, so groups g + 2 and g + 3 are requested right behind
// them.  nk points at -k of group g - OFF (OFF is a compile-time offset, so the reads use immediate offsets).
template <bool PRIMARY, bool K_LDS, int OFF, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void pair_step(geom_ptr geom, int g, nk_lds_ptr nk, GroupRec& ga, GroupRec& gb, f4v& ka,
                                                 f4v& kb, ExactPtr exact, CandList<SHORT>& cand, const ScanRay& r, f3 o, f3 u,
                                                 float tmin, int prev, const PrimaryCtx& pc RTOW_PROF_PARAM) {
  const f2v a0 = pair_filter<!K_LDS>(ga.a.s01, ga.a.s23, ga.a.s45, ka.xy, r);
  const f2v a1 = pair_filter<!K_LDS>(ga.a.s67, ga.b.xy, ga.b.zw, ka.zw, r);
  ga = load_group<K_LDS, false>(geom, g + 2, ka, nullptr);
  if (K_LDS) ka = nk[OFF + 2];
  const f2v b0 = pair_filter<!K_LDS>(gb.a.s01, gb.a.s23, gb.a.s45, kb.xy, r);
  const f2v b1 = pair_filter<!K_LDS>(gb.a.s67, gb.b.xy, gb.b.zw, kb.zw, r);
  // the next -k may only be requested once this one has been consumed: hoisted above the last FMA it would need
  // registers of its own, and the trip would begin by copying four registers
  __builtin_amdgcn_sched_barrier(0);
  gb = load_group<K_LDS, false>(geom, g + 3, kb, nullptr);
  if (K_LDS) kb = nk[OFF + 3];
  const float ma = fmaxf(fmaxf(a0.x, a0.y), fmaxf(a1.x, a1.y));
  const float mx = fmaxf(fmaxf(b0.x, b0.y), fmaxf(fmaxf(b1.x, b1.y), ma));
  if (__ballot(mx >= r.ozt.y) != 0) {   // wave-uniform: the list's fill count stays a scalar
    RTOW_COUNT(9, 1);
    const int j = 4 * g;
    if (__ballot(ma >= r.ozt.y) != 0) {
      cand_push(cand, a0.x >= r.ozt.y, j);
      cand_push(cand, a0.y >= r.ozt.y, j + 1);
      cand_push(cand, a1.x >= r.ozt.y, j + 2);
      cand_push(cand, a1.y >= r.ozt.y, j + 3);
      if (cand.n > ListEntry<SHORT>::kEntries - 256u) cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
    }
    cand_push(cand, b0.x >= r.ozt.y, j + 4);
    cand_push(cand, b0.y >= r.ozt.y, j + 5);
    cand_push(cand, b1.x >= r.ozt.y, j + 6);
    cand_push(cand, b1.y >= r.ozt.y, j + 7);
    if (cand.n > ListEntry<SHORT>::kEntries - 256u) cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
  }
}

// Tables beyond kMaxSpheresKInLds entries: -k does not fit LDS as a whole, and taken from scalar registers it costs a
// packed add per pair of spheres (pair_filter<K_SGPR>).  Each WAVE therefore streams -k through a small ring of its
// own in LDS: tiles of kTileGroups groups (256 spheres, 1 KB), written by global_load_lds - a load that goes from
// memory to LDS without passing through registers, 16 bytes per lane - one tile ahead of the scan, so
// that the filter can take -k as the addend of its first FMA exactly as small tables do.  A tile costs one load and a
// handful of instructions against 128 packed adds.  The ring is private to the wave: no barrier, just the wave's own
// wait for its load.  The centres keep coming through scalar registers, as always.
constexpr int kTileGroups = 64;
constexpr int kRingF4 = 2 * kTileGroups + 2;   // two tiles + the two groups a trip reads ahead past a tile's end

template <bool PRIMARY, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_tiled(geom_ptr geom, ExactPtr exact, const f4v* ring, CandList<SHORT>& cand,
                                                  int n_groups, const ScanRay& r, f3 o, f3 u, float tmin, int prev,
                                                  const PrimaryCtx& pc RTOW_PROF_PARAM) {
  typedef const float __attribute__((address_space(1)))* global_f32_ptr;
  typedef f4v __attribute__((address_space(3)))* ring_ptr;
  const global_f32_ptr nk_global = (global_f32_ptr)(uintptr_t)KARG(geom_nk);   // -k per list entry, padded to whole tiles + one
  const ring_ptr ring_lds = (ring_ptr)(f4v*)ring;
  const int n_tiles = (n_groups + kTileGroups - 1) / kTileGroups;
  // tile t -> half (t & 1) of the ring; the data is in LDS once the wave's VMEM counter has drained
  // (one load of 16 bytes by all 64 lanes: a branch on the lane number around a narrower load makes the compiler treat
  // the whole tile loop as divergent, and every such load costs the wave some hundred cycles to issue whatever its size)
  auto request = [&](int t) {
    __builtin_amdgcn_global_load_lds(nk_global + (size_t)t * (4 * kTileGroups) + 4 * cand.lane, ring_lds + (t & 1) * kTileGroups, 16, 0, 0);
  };
  request(0);
  f4v ka = {0, 0, 0, 0}, kb = ka, unused = ka;
  GroupRec ga = load_group<true, false>(geom, 0, unused, nullptr);   // the first 12 floats of a record: the centres
  GroupRec gb = load_group<true, false>(geom, 1, unused, nullptr);
  for (int t = 0; t < n_tiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile t has landed (and whatever the exact tests had in flight)
    if (t + 1 < n_tiles) request(t + 1);               // into the half the scan of tile t - 1 has finished with
    // the LDS address of the tile's -k in a VGPR, so that the reads of a trip use immediate offsets (see scan_regular)
    uint32_t nk_addr = (uint32_t)(uintptr_t)(nk_lds_ptr)(ring + (t & 1) * kTileGroups);
    asm volatile("" : "+v"(nk_addr));
    nk_lds_ptr nk = (nk_lds_ptr)(uintptr_t)nk_addr;
    ka = nk[0];   // a trip ends by reading two groups past itself: at a tile's end that was not this tile's data
    kb = nk[1];
    const int end = min(n_groups, (t + 1) * kTileGroups);
    int g = t * kTileGroups;
    for (; g + 8 <= end; g += 8) {
      pair_step<PRIMARY, true, 0>(geom, g, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, true, 2>(geom, g + 2, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, true, 4>(geom, g + 4, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, true, 6>(geom, g + 6, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      nk += 8;
    }
    for (; g < end; g += 2) {
      pair_step<PRIMARY, true, 0>(geom, g, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      nk += 2;
    }
  }
  cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
}

// n_groups is even and the table holds two more (all-padding) groups after the last one, so the
// loop can always request up to three groups ahead without a bounds test.
//
// K_LDS: -k of every sphere sits in LDS (copied once per workgroup) and reaches the filter as a
// VGPR pair through a broadcast ds_read_b128 issued one group ahead.  That lets the second chain
// start as fma(cx, 2o.x, -k): a VALU instruction can read only one SGPR operand, so with k in an
// SGPR the subtraction would cost a packed instruction of its own.  Tables too large for LDS use
// the SGPR copy of k (moved to VGPRs first; same arithmetic, two more instructions per group).

// Called by all 64 lanes of the wave; lanes without a ray pass live = false and offer nothing.  The
// caller has put every lane's closest hit so far into cand.best (and pc.best_id for a primary scan) and
// reads the result back from there.
template <bool PRIMARY, bool K_LDS, int AXIS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_regular(geom_ptr geom, ExactPtr exact, const f4v* nk_lds,
                                                    CandList<SHORT>& cand, int n_groups, float cmax, bool live, f3 o, f3 u,
                                                    float tmin, int prev, const PrimaryCtx& pc, SlabRay sr RTOW_PROF_PARAM) {
  static_assert(AXIS < 0 || PRIMARY, "the slab test is for the camera rays");
  if (n_groups <= 0) return;
  constexpr bool SLAB = AXIS >= 0;   // the slab test runs where the wave's rays are one pixel's camera rays
  const slab_ptr slab = (SLAB && !K_LDS) ? (slab_ptr)(uintptr_t)KARG(geom_slab) : nullptr;
  ScanRay r = scan_ray(o, u, cmax);
  if (!live) { r.ozt.y = INFINITY; sr.pre.x = INFINITY; }   // an infinite value passes neither test
  // The LDS address of -k lives in a VGPR that advances by two groups per trip; the reads then use
  // immediate offsets.  (Left to itself the compiler keeps the address in an SGPR and copies it to
  // a VGPR for every read: one more VALU instruction per group.)
  uint32_t nk_addr = K_LDS ? (uint32_t)(uintptr_t)(nk_lds_ptr)nk_lds : 0u;
  asm volatile("" : "+v"(nk_addr));
  nk_lds_ptr nk = (nk_lds_ptr)(uintptr_t)nk_addr;
  // with the slab test in front, one group in eleven needs its -k: it is read then, not one group ahead
  constexpr bool NK_LATE = SLAB && K_LDS;
  f4v ka = (K_LDS && !NK_LATE) ? nk[0] : f4v{0, 0, 0, 0}, kb = ka;
  GroupRec ga = load_group<K_LDS, SLAB>(geom, 0, ka, slab);
#ifndef RTOW_NO_TILED_K
  if constexpr (!SLAB && !K_LDS) {   // nk_lds is this wave's ring here (see scan_tiled)
    scan_tiled<PRIMARY>(geom, exact, nk_lds, cand, n_groups, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
    return;
  }
#endif
#ifndef RTOW_NO_PAIRED_GROUPS
  if (!SLAB) {
    // Scans without the slab test decide about EIGHT spheres with one compare and one branch (pair_step), and the
    // loop is unrolled four pairs deep so that -k of all eight groups is read through immediate offsets from one
    // address register: the address advances once per 32 spheres instead of once per eight.
    GroupRec gb = load_group<K_LDS, SLAB>(geom, 1, kb, slab);
    if (K_LDS) kb = nk[1];
    int g = 0;
    for (; g + 8 <= n_groups; g += 8) {
      pair_step<PRIMARY, K_LDS, 0>(geom, g, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, K_LDS, 2>(geom, g + 2, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, K_LDS, 4>(geom, g + 4, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      pair_step<PRIMARY, K_LDS, 6>(geom, g + 6, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      if (K_LDS) nk += 8;
    }
    for (; g < n_groups; g += 2) {
      pair_step<PRIMARY, K_LDS, 0>(geom, g, nk, ga, gb, ka, kb, exact, cand, r, o, u, tmin, prev, pc RTOW_PROF_ARG);
      if (K_LDS) nk += 2;
    }
    cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
    return;
  }
#endif
  // (NK_LATE: the address of a group's -k is formed from the group index where it is needed - one group in eleven -
  // instead of being carried along in a register that every trip has to advance)
  const nk_lds_ptr nk_base = (nk_lds_ptr)nk_lds;
#ifndef RTOW_SLAB_LOADS_UNORDERED
  if (SLAB) {
    // A wait for scalar loads is a wait for ALL of them (they return out of order), so a group requested just before
    // the wait for its predecessor is waited for at once, and with seven instructions per group there is little
    // other work to hide that behind.  The empty asm statements read a register of the group about to be used: the
    // wait lands there, and the next group is requested only after it - every request then has one group's worth of
    // work (of this wave and of the four others of the SIMD) to arrive in.
    for (int g = 0; g < n_groups; g += 2) {
      asm volatile("" : : "s"(ga.a.s0));
      __builtin_amdgcn_sched_barrier(0);
      const GroupRec gb = load_group<K_LDS, SLAB>(geom, g + 1, kb, slab);
      __builtin_amdgcn_sched_barrier(0);
      group_test<PRIMARY, AXIS, NK_LATE, !K_LDS>(ga, ka, NK_LATE ? nk_base + g : nk, 4 * g, exact, cand, r, sr, o, u, tmin, prev, pc RTOW_PROF_ARG);
      asm volatile("" : : "s"(gb.a.s0));
      __builtin_amdgcn_sched_barrier(0);
      ga = load_group<K_LDS, SLAB>(geom, g + 2, ka, slab);
      __builtin_amdgcn_sched_barrier(0);
      group_test<PRIMARY, AXIS, NK_LATE, !K_LDS>(gb, kb, NK_LATE ? nk_base + g + 1 : nk + 1, 4 * g + 4, exact, cand, r, sr, o, u, tmin, prev, pc RTOW_PROF_ARG);
    }
    cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
    return;
  }
#endif
  for (int g = 0; g < n_groups; g += 2) {
    const GroupRec gb = load_group<K_LDS, SLAB>(geom, g + 1, kb, slab);
    if (K_LDS && !NK_LATE) kb = nk[1];
    group_test<PRIMARY, AXIS, NK_LATE, !K_LDS>(ga, ka, NK_LATE ? nk_base + g : nk, 4 * g, exact, cand, r, sr, o, u, tmin, prev, pc RTOW_PROF_ARG);
    ga = load_group<K_LDS, SLAB>(geom, g + 2, ka, slab);
    if (K_LDS && !NK_LATE) ka = nk[2];
    group_test<PRIMARY, AXIS, NK_LATE, !K_LDS>(gb, kb, NK_LATE ? nk_base + g + 1 : nk + 1, 4 * g + 4, exact, cand, r, sr, o, u, tmin, prev, pc RTOW_PROF_ARG);
    if (K_LDS && !NK_LATE) nk += 2;
  }
  cand_drain<PRIMARY>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
}

// Closest hit of the camera rays of a PRIMARY iteration when those rays form a narrow bundle (64 samples of one
// pixel through a small lens, or of a few neighbouring pixels).  The wave first walks the table with the SPHERES
// spread over its lanes and tests each against the bundle as a whole - one axis ray (a, w) and the spread of the 64
// rays about it - and only the groups with a sphere near the bundle go through the per-ray filter and the exact
// tests.  The list stays a plain list, walked from the first entry to the last for every bundle; no structure is
// kept over the scene or between iterations.
//
// Bound.  Let delta_o >= |o_i - a| and delta_u >= |u_i - w| for every ray i of the wave (unit directions).  The point
// of ray i nearest to a centre c is o_i + s u_i with |s| <= |c - o_i| <= |c - a| + delta_o, and that point lies within
// delta_o + |s| delta_u of the axis point a + s w.  A sphere that ray i's exact test accepts has its centre within |r|
// of the ray's line, hence within
//     |r| + rho(c),   rho(c) = delta_o + (|c - a| + delta_o) delta_u
// of the axis line.  The square root is avoided with |c - a| <= q / (2 L) + L / 2 for q = |c - a|^2 and any L > 0
// (tight at |c - a| = L; L = the camera's distance from the world origin, floored at 1):
//     rho(c) <= A + B q,   A = delta_o (1 + delta_u) + delta_u L / 2,   B = delta_u / (2 L).
// A group goes on unless, for all four of its spheres,  q - ((c - a).w)^2  >  (|r| + A + B q)^2 + slack.
// Roundings: the spreads are measured on the binary32 rays the filter uses, which lie within 2^-24 |o|_1 and 2^-21
// of the binary64 rays; the table's centres within 2^-24 |c|_1 of the binary64 centres; the left side is evaluated
// with an absolute error below 2^-19 q.  delta_o, delta_u are therefore enlarged by 2^-10 of themselves plus
// 2^-20 (|a|_1 + cmax + 1) and 2^-20, |r| (a raw v_sqrt_f32 of the table's r^2) by 2^-16 of itself, and slack is
// 2^-18 q.  A NaN anywhere lets the group through (the comparison is written so), where the filter decides as before.
// Entries that must never pass (padding, spheres tested elsewhere) have an all-zero record here and k = +inf in the
// filter: if the bundle test lets one through, the filter stops it.
//
// How far the bundle test thins the table depends on the rays: one pixel's samples through the book camera's lens
// leave a handful of groups; a wave whose lanes belong to many pixels, or a wide lens, leaves more, and a wave that
// straddles the end of an image row leaves nearly all of them - the scan then costs the bundle test on top of a
// plain scan, and returns the same hits.
template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_bundle(geom_ptr geom, ExactPtr exact, const f4v* nk_lds, CandList<SHORT>& cand,
                                                   int n_groups, float cmax, bool live, f3 o, f3 u, f3 a, f3 w,
                                                   const PrimaryCtx& pc RTOW_PROF_PARAM) {
  if (n_groups <= 0) return;
  ScanRay r = scan_ray(o, u, cmax);
  if (!live) r.ozt.y = INFINITY;
  // spread of the wave's rays about the axis: two wave-wide maxima of non-negative numbers, taken as integers
  // (the bit patterns order like the values, NaN above them all) with LDS atomics on two words of the empty list
  float A, B;
  {
    const f3 eo = {o.x - a.x, o.y - a.y, o.z - a.z}, eu = {u.x - w.x, u.y - w.y, u.z - w.z};
    const lds_u32_ptr red = (lds_u32_ptr)(uint32_t*)cand.list;
    if (cand.lane == 0) { red[0] = 0u; red[1] = 0u; }
    lds_order();
    __hip_atomic_fetch_max(red, __float_as_uint(dot3(eo, eo)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    __hip_atomic_fetch_max(red + 1, __float_as_uint(dot3(eu, eu)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    lds_order();
    const float mo = __uint_as_float(red[0]), mu = __uint_as_float(red[1]);
    lds_order();   // read before the list is used again
    const float delta_o = fmaf(__builtin_amdgcn_sqrtf(mo), 1.0f + 0x1p-10f, 0x1p-20f * (fabsf(a.x) + fabsf(a.y) + fabsf(a.z) + cmax + 1.0f));
    const float delta_u = fmaf(__builtin_amdgcn_sqrtf(mu), 1.0f + 0x1p-10f, 0x1p-20f);
    const float L = __builtin_amdgcn_sqrtf(fmaf(a.x, a.x, fmaf(a.y, a.y, fmaf(a.z, a.z, 1.0f))));
    A = (delta_o * (1.0f + delta_u) + 0.5f * delta_u * L) * (1.0f + 0x1p-10f);
    B = 0.5f * delta_u * __builtin_amdgcn_rcpf(L) * (1.0f + 0x1p-10f);
  }
  const nk_lds_ptr nk = (nk_lds_ptr)nk_lds;
  for (int g0 = 0; g0 < n_groups; g0 += 64) {
    const int g = g0 + (int)cand.lane;
    bool near = false;
    if (g < n_groups) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const f4v sp = exact[4 * g + i];   // (cx, cy, cz, r^2)
        const float ocx = sp.x - a.x, ocy = sp.y - a.y, ocz = sp.z - a.z;
        float q = ocx * ocx; q = fmaf(ocy, ocy, q); q = fmaf(ocz, ocz, q);
        float b = ocx * w.x; b = fmaf(ocy, w.y, b); b = fmaf(ocz, w.z, b);
        const float dist2 = fmaf(-b, b, q);
        const float reach = fmaf(__builtin_amdgcn_sqrtf(sp.w), 1.0f + 0x1p-16f, fmaf(B, q, A));
        const float rhs = fmaf(q, 0x1p-18f, reach * reach);
        near = near || !(dist2 > rhs);
      }
    }
    uint64_t m = __ballot(near);
    while (m != 0) {
      const int gg = g0 + (int)__builtin_ctzll(m);
      m &= m - 1;
      RTOW_COUNT(24, 1);
      f4v ka = {0, 0, 0, 0};
      const GroupRec grp = load_group<K_LDS, false>(geom, gg, ka, nullptr);
      if (K_LDS) ka = nk[gg];
      group_test<true, -1, false, !K_LDS>(grp, ka, nk, 4 * gg, exact, cand, r, SlabRay{f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}}, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
    }
  }
  cand_drain<true>(cand, exact, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
}

// Closest hit of the camera rays of a PRIMARY iteration behind the interval test.  For every sphere the host has
// worked out, from scene and camera, the range of the image coordinate s - the first argument of getRay
// (camera.nim:34), which runs across the image - outside which no camera ray can meet it, whatever its t and its point
// on the lens (rtow_capi.hip, refresh_spans).  Every ray compares its own s with that range for every sphere of the
// list, in list order: one packed subtraction per two spheres, |s - middle| of a group's four against the largest half
// width among them (one scalar per group), five instructions per group where the filter takes seventeen.  The groups
// in which some lane's ray passes see the same test for t, the coordinate up the image, and only those that pass
// both go through the filter and the exact tests, as in a scan without any such test.  What the tests drop are
// spheres the ray cannot meet, so hits, images and segment counts are those of the plain scan.
typedef const f4v __attribute__((address_space(4)))* span_mid_ptr;
typedef const float __attribute__((address_space(4)))* span_half_ptr;

// middle - coordinate for a group's four spheres (one half of cv - HI selects which - for both lanes of a packed
// subtraction), and the smallest of the four distances
template <bool HI>
static __device__ __forceinline__ float interval_distance(const f4v mid, const f2v cv) {
  f2v d0, d1;
  if (HI) {
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "s"(mid.xy), "v"(cv));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "s"(mid.zw), "v"(cv));
  } else {
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "s"(mid.xy), "v"(cv));
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "s"(mid.zw), "v"(cv));
  }
  return min_abs4(d0, d1);
}

// st = (s, t) of the lane's ray
template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void interval_group(const f4v mid, const float half, const f2v st, int g, int n_groups, geom_ptr geom,
                                                      nk_lds_ptr nk, ExactPtr exact, CandList<SHORT>& cand, const ScanRay& r,
                                                      f3 o, f3 u, const PrimaryCtx& pc RTOW_PROF_PARAM) {
  if (__ballot(interval_distance<false>(mid, st) <= half) == 0) return;   // wave-uniform: no ray of the wave can meet any of the four
  RTOW_COUNT(25, 1);
  // One group in eleven gets here.  Its ranges of t and its centres are requested together and waited for once; the
  // same test along the image's other axis then lets about one in five of these go on to the filter.
  const span_mid_ptr mids_t = (span_mid_ptr)(uintptr_t)KARG(span_mid) + (n_groups + kSpanPadGroups);
  const span_half_ptr halves_t = (span_half_ptr)(uintptr_t)KARG(span_half) + (n_groups + kSpanPadGroups);
  const f4v mid_t = mids_t[g];
  const float half_t = halves_t[g];
  f4v ka = {0, 0, 0, 0};
  const GroupRec grp = load_group<K_LDS, false>(geom, g, ka, nullptr);
  if (K_LDS) ka = nk[g];
  if (__ballot(interval_distance<true>(mid_t, st) <= half_t) == 0) return;
  RTOW_COUNT(24, 1);
  group_test<true, -1, false, !K_LDS>(grp, ka, nk, 4 * g, exact, cand, r, SlabRay{f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}}, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
}

template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_interval(geom_ptr geom, ExactPtr exact, const f4v* nk_lds, CandList<SHORT>& cand,
                                                     int n_groups, float cmax, bool live, f3 o, f3 u, float s, float t,
                                                     const PrimaryCtx& pc RTOW_PROF_PARAM) {
  if (n_groups <= 0) return;
  ScanRay r = scan_ray(o, u, cmax);
  if (!live) { r.ozt.y = INFINITY; s = INFINITY; t = INFINITY; }   // an infinite distance passes no finite half width, an infinite filter threshold nothing
  const f2v st = {s, t};
  const nk_lds_ptr nk = (nk_lds_ptr)nk_lds;
  // Two groups per scalar load: the middles of eight spheres and two half widths.  A wait for scalar loads is a wait
  // for all of them (they return out of order), so the request for the next pair is issued right AFTER the wait for
  // the current one - the empty asm statements read a register of the pair about to be used, which is where the wait
  // lands - and then has two groups' worth of work (of this wave and of the four others of the SIMD) to arrive in.
  // The tables hold four all-padding groups behind the last one.
  typedef const f8v __attribute__((address_space(4)))* mid_pair_ptr;
  typedef const f2v __attribute__((address_space(4)))* half_pair_ptr;
  const mid_pair_ptr mids = (mid_pair_ptr)(uintptr_t)KARG(span_mid);
  const half_pair_ptr halves = (half_pair_ptr)(uintptr_t)KARG(span_half);
  f8v ma = mids[0];
  f2v ha = halves[0];
  for (int g = 0; g < n_groups; g += 4) {
    asm volatile("" : : "s"(ma.s0));
    __builtin_amdgcn_sched_barrier(0);
    const f8v mb = mids[g / 2 + 1];
    const f2v hb = halves[g / 2 + 1];
    __builtin_amdgcn_sched_barrier(0);
    interval_group<K_LDS>(ma.lo, ha.x, st, g, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    interval_group<K_LDS>(ma.hi, ha.y, st, g + 1, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    asm volatile("" : : "s"(mb.s0));
    __builtin_amdgcn_sched_barrier(0);
    ma = mids[g / 2 + 2];
    ha = halves[g / 2 + 2];
    __builtin_amdgcn_sched_barrier(0);
    // (when the number of groups is no multiple of four these two are padding: they pass nothing)
    interval_group<K_LDS>(mb.lo, hb.x, st, g + 2, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    interval_group<K_LDS>(mb.hi, hb.y, st, g + 3, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
  }
  cand_drain<true>(cand, exact, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
}

// Closest hit of the camera rays of a PRIMARY iteration behind the mask test.  The image is cut into up to 32 tiles
// (columns along s, rows along t; the host picks the grid, rtow_capi.hip, refresh_spans), and for every group of four
// spheres the host has marked, in one 32-bit word, the tiles in which a camera ray can meet one of the four: the tiles
// the ranges of the interval test reach into.  Every ray works out the bit of the tile it aims at, once, and tests it
// against the word of every group of the list, in list order; the groups in which some lane's ray finds its bit set
// see the interval test (both axes), and those that pass it the filter and the exact tests.  The words of four groups
// are tested with one comparison (their ANDs with the ray's bit ORed together), so a group costs a ray about half an
// instruction where the interval test takes five.  Like the interval test this drops only spheres the ray cannot
// meet: a mask holds every tile its spheres' ranges touch, and the tile of a ray is a monotonic function of s and t
// evaluated in the same binary32 operations here and on the host.
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
typedef uint32_t u8v __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ uint32_t tile_bit(float s, float t) {
  const float xs = __builtin_amdgcn_fmed3f(__builtin_fmaf(s, KARG(tile_ks), KARG(tile_cs)), 0.0f, KARG(tile_smax));
  const float xt = __builtin_amdgcn_fmed3f(__builtin_fmaf(t, KARG(tile_kt), KARG(tile_ct)), 0.0f, KARG(tile_tmax));
  return 1u << ((uint32_t)xs + KARG(tile_ns) * (uint32_t)xt);
}

// one group that passed the mask test: its ranges of s and t, its centres and -k are requested together and waited
// for once
template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void masked_group(const f2v st, int g, int n_groups, geom_ptr geom, nk_lds_ptr nk, ExactPtr exact,
                                                    CandList<SHORT>& cand, const ScanRay& r, f3 o, f3 u, const PrimaryCtx& pc RTOW_PROF_PARAM) {
  RTOW_COUNT(26, 1);
  const span_mid_ptr mids_s = (span_mid_ptr)(uintptr_t)KARG(span_mid);
  const span_half_ptr halves_s = (span_half_ptr)(uintptr_t)KARG(span_half);
  const f4v mid_s = mids_s[g], mid_t = mids_s[g + n_groups + kSpanPadGroups];
  const float half_s = halves_s[g], half_t = halves_s[g + n_groups + kSpanPadGroups];
  f4v ka = {0, 0, 0, 0};
  const GroupRec grp = load_group<K_LDS, false>(geom, g, ka, nullptr);
  if (K_LDS) ka = nk[g];
  asm volatile("" : : "s"(mid_t.x), "s"(half_t), "s"(grp.a.s0), "s"(grp.b.x));   // all of them issued before the first is waited for
  if (__ballot(interval_distance<false>(mid_s, st) <= half_s) == 0) return;
  RTOW_COUNT(25, 1);
  if (__ballot(interval_distance<true>(mid_t, st) <= half_t) == 0) return;
  RTOW_COUNT(24, 1);
  group_test<true, -1, false, !K_LDS>(grp, ka, nk, 4 * g, exact, cand, r, SlabRay{f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}}, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
}

template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void masked_block(const u4v m, const uint32_t bit, const f2v st, int g, int n_groups, geom_ptr geom,
                                                    nk_lds_ptr nk, ExactPtr exact, CandList<SHORT>& cand, const ScanRay& r, f3 o, f3 u,
                                                    const PrimaryCtx& pc RTOW_PROF_PARAM) {
  // the ray's bit against the word of each of the four groups, the results ORed together: v_and_b32 and three
  // v_and_or_b32 (spelled out: from (a & x) | (b & x) ... the compiler builds a longer sequence)
  uint32_t any;
  asm("v_and_b32 %0, %1, %5\n\tv_and_or_b32 %0, %2, %5, %0\n\tv_and_or_b32 %0, %3, %5, %0\n\tv_and_or_b32 %0, %4, %5, %0"
      : "=&v"(any) : "s"(m.x), "s"(m.y), "s"(m.z), "s"(m.w), "v"(bit));
  if (__ballot(any != 0u) == 0) return;   // wave-uniform: no ray of the wave aims at a tile any of the sixteen reaches into
  if (__ballot((m.x & bit) != 0u) != 0) masked_group<K_LDS>(st, g, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
  if (__ballot((m.y & bit) != 0u) != 0) masked_group<K_LDS>(st, g + 1, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
  if (__ballot((m.z & bit) != 0u) != 0) masked_group<K_LDS>(st, g + 2, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
  if (__ballot((m.w & bit) != 0u) != 0) masked_group<K_LDS>(st, g + 3, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
}

template <bool K_LDS, bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_masked(geom_ptr geom, ExactPtr exact, const f4v* nk_lds, CandList<SHORT>& cand,
                                                   int n_groups, float cmax, bool live, f3 o, f3 u, float s, float t,
                                                   const PrimaryCtx& pc RTOW_PROF_PARAM) {
  if (n_groups <= 0) return;
  ScanRay r = scan_ray(o, u, cmax);
  uint32_t bit = tile_bit(s, t);
  if (!live) { r.ozt.y = INFINITY; s = INFINITY; t = INFINITY; bit = 0u; }   // no bit: passes no mask
  const f2v st = {s, t};
  const nk_lds_ptr nk = (nk_lds_ptr)nk_lds;
  // eight masks per scalar load, requested one load ahead (see scan_interval for the ordering); the table ends in
  // kTileMaskPad zero words
  typedef const u8v __attribute__((address_space(4)))* mask8_ptr;
  const mask8_ptr masks = (mask8_ptr)(uintptr_t)KARG(tile_mask);
  u8v ma = masks[0];
  for (int g = 0; g < n_groups; g += 16) {
    asm volatile("" : : "s"(ma.s0));
    __builtin_amdgcn_sched_barrier(0);
    const u8v mb = masks[g / 8 + 1];
    __builtin_amdgcn_sched_barrier(0);
    masked_block<K_LDS>(ma.lo, bit, st, g, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    masked_block<K_LDS>(ma.hi, bit, st, g + 4, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    asm volatile("" : : "s"(mb.s0));
    __builtin_amdgcn_sched_barrier(0);
    ma = masks[g / 8 + 2];
    __builtin_amdgcn_sched_barrier(0);
    masked_block<K_LDS>(mb.lo, bit, st, g + 8, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
    masked_block<K_LDS>(mb.hi, bit, st, g + 12, n_groups, geom, nk, exact, cand, r, o, u, pc RTOW_PROF_ARG);
  }
  cand_drain<true>(cand, exact, o, u, 0.0f, -1, pc RTOW_PROF_ARG);
}

// The same closest-hit query for a wave that has only a few paths left (the end of a frame, when no
// work is left to refill the idle lanes): instead of every lane walking the whole table for its own
// ray, the rays are taken one at a time and the TABLE is spread over the 64 lanes - lane L screens
// spheres L, L + 64, ... for the ray of the current owner, read from the records in LDS.  The screen
// is the same arithmetic (filter_value is pair_filter one sphere at a time), survivors go to the same
// list with the owner's lane number, and the exact tests are shared as always, so the result is the
// same hit; a wave with m paths pays about m * 100 instructions instead of 2 000.
// Needs -k and the exact-test records in LDS (tables up to kMaxSpheresExactInLds entries).
constexpr uint32_t kSparseLanes = 12;

template <bool SHORT, typename ExactPtr>
static __device__ __forceinline__ void scan_sparse(ExactPtr exact, const f4v* nk_lds, CandList<SHORT>& cand, int n_pad,
                                                   float cmax, bool live, f3 o, f3 u, float tmin, int prev,
                                                   const PrimaryCtx& pc RTOW_PROF_PARAM) {
  typedef const float __attribute__((address_space(3)))* lds_f32_ptr;
  const lds_f32_ptr nk = (lds_f32_ptr)(const float*)nk_lds;
  uint64_t todo = __ballot(live);
  while (todo != 0) {
    const int owner = __builtin_ctzll(todo);
    todo &= todo - 1;
    const int ob = owner << 2;
    const f3 oo = {from_lane(ob, o.x), from_lane(ob, o.y), from_lane(ob, o.z)};
    const f3 uu = {from_lane(ob, u.x), from_lane(ob, u.y), from_lane(ob, u.z)};
    const FilterRay fr = filter_ray(oo, uu, cmax);
    for (int j0 = 0; j0 < n_pad; j0 += 64) {
      const int j = j0 + (int)cand.lane;
      bool pass = false;
      if (j < n_pad) {
        const f4v sp = exact[j];
        pass = filter_value(sp.x, sp.y, sp.z, -nk[j], uu, fr) >= fr.threshold;   // padding and binary64 spheres: k = +inf
      }
      const uint64_t m = __ballot(pass);
      if (m != 0) {
        if (pass) cand.list[cand.n + lanes_below(m)] = (typename CandList<SHORT>::entry_t)(((uint32_t)owner << ListEntry<SHORT>::kShift) | (uint32_t)j);
        cand.n += (uint32_t)__popcll(m);
        if (cand.n > ListEntry<SHORT>::kEntries - 64u) cand_drain<false>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
      }
    }
  }
  cand_drain<false>(cand, exact, o, u, tmin, prev, pc RTOW_PROF_ARG);
}

// A path that ends with light (the sky) adds its fixed-point radiance to its pixel.  The paths of a wave
// belong to a handful of pixels at a time (a chunk is a run of consecutive samples), so the sums are
// collected in a small per-wave cache in LDS first - four slots of three 64-bit words,
// filled with LDS atomics - and a slot goes to the frame (three 64-bit global atomics) when another
// pixel claims it or the wave ends: one set of global atomics per pixel and wave.  Integer sums do not
// depend on the order they arrive in.  (Sums kept per lane in registers and handed over when the lane
// changes pixel cost six registers per lane in a kernel that spills; a shuffle reduction over the
// lanes of a pixel costs ~100 VALU instructions per round; the LDS atomics cost the VALU one each.)
constexpr uint32_t kSumSlots = 4;
constexpr uint32_t kSumCacheDwords = 32;   // per wave: kSumSlots * 3 words of 8 bytes, padded to 128 bytes
constexpr uint32_t kNoPixel = 0xffffffffu;

// The tags are wave-uniform scalars on purpose.  (Kept in the lanes of a vector register they are only
// safe while that register is never spilled inside a divergent region: a spill stores the active lanes
// only, and the lanes that hold the tags need not be among them.)
struct SumCache {
  lds_u64_ptr words;             // this wave's kSumSlots * 3 words
  uint32_t tag[kSumSlots];       // the pixel each slot holds
  uint32_t victim;               // the slot the next new pixel takes (round robin)
#ifdef RTOW_GUARD
  long long n_pixels;
#endif
};
static_assert(kSumSlots == 4, "tag_of / set_tag select among four scalars");

static __device__ __forceinline__ uint32_t tag_of(const SumCache& c, uint32_t slot) {
  return slot == 0u ? c.tag[0] : slot == 1u ? c.tag[1] : slot == 2u ? c.tag[2] : c.tag[3];
}
// the slot that holds `pix`, or kSumSlots
static __device__ __forceinline__ uint32_t slot_of(const SumCache& c, uint32_t pix) {
  return c.tag[0] == pix ? 0u : c.tag[1] == pix ? 1u : c.tag[2] == pix ? 2u : c.tag[3] == pix ? 3u : kSumSlots;
}
static __device__ __forceinline__ void set_tag(SumCache& c, uint32_t slot, uint32_t pix) {
  c.tag[0] = slot == 0u ? pix : c.tag[0]; c.tag[1] = slot == 1u ? pix : c.tag[1];
  c.tag[2] = slot == 2u ? pix : c.tag[2]; c.tag[3] = slot == 3u ? pix : c.tag[3];
}

static __device__ __forceinline__ void sum_cache_evict(const SumCache& c, uint32_t slot, uint32_t pix, uint32_t lane) {
  unsigned long long* const accum = KARG(accum);
  // LDS serves one wave's requests in order: the adds of earlier instructions are in the words read here
  lds_order();
#ifdef RTOW_PROFILE_COUNTERS
  if (lane == 0u) atomicAdd(&g_prof[0], 1ull);
#endif
  if (lane < 3u) {
    const unsigned long long v = c.words[3u * slot + lane];
    c.words[3u * slot + lane] = 0ull;
    if (RTOW_GUARDED(1, pix, c.n_pixels)) atomicAdd(&accum[3ull * pix + lane], v);
  }
  lds_order();
}

// Must be called by all 64 lanes together; lanes with `want` set add (acc_r, acc_g, acc_b), in sum units, to pixel accpix.
static __device__ __forceinline__ void flush_sums(SumCache& c, bool want, uint32_t accpix,
                                                  uint64_t acc_r, uint64_t acc_g, uint64_t acc_b, uint32_t lane) {
  uint64_t todo = __ballot(want);
  while (todo != 0) {
    const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
    const uint32_t pix = (uint32_t)__builtin_amdgcn_readlane((int)accpix, (int)leader);
    // The cache is fully associative: a wave works on the pixels of its current chunk and of a few paths
    // left over from earlier chunks, whose pixel numbers are unrelated - mapped directly, two of them on
    // one slot evict each other with every sample.
    uint32_t slot = slot_of(c, pix);
    if (slot == kSumSlots) {
      slot = c.victim;
      c.victim = (c.victim + 1u) & (kSumSlots - 1u);
      const uint32_t held = tag_of(c, slot);
      if (held != kNoPixel) sum_cache_evict(c, slot, held, lane);
      set_tag(c, slot, pix);
    }
    const bool mine = want && accpix == pix;
    if (mine) {
      __hip_atomic_fetch_add(c.words + 3u * slot + 0u, (unsigned long long)acc_r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(c.words + 3u * slot + 1u, (unsigned long long)acc_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(c.words + 3u * slot + 2u, (unsigned long long)acc_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    todo &= ~__ballot(mine);
  }
}

static __device__ __forceinline__ void sum_cache_drain(SumCache& c, uint32_t lane) {
  for (uint32_t slot = 0; slot < kSumSlots; ++slot) {
    const uint32_t held = c.tag[slot];
    if (held != kNoPixel) sum_cache_evict(c, slot, held, lane);
    c.tag[slot] = kNoPixel;
  }
}

// fields of a pool entry: a path whose primary segment hit something and scattered
enum : uint32_t { kPoolO = 0, kPoolD = 3, kPoolT = 6, kPoolX0 = 9, kPoolKey1 = 10, kPoolPix = 11, kPoolPrev = 12 };
static_assert(kPoolFields >= 14, "the pool region also holds seven doubles per lane during a primary scan");

// The wave alternates between two kinds of iteration, each one sphere scan at (nearly) full lanes:
//
//  PRIMARY  when lanes are idle and the pool is empty: the next (up to) 64 paths of the chunk are started
//           together, one per lane.  Camera rays (raytracer.nim:41-43, camera.nim:34-39) and their closest hit
//           are evaluated in binary64 exactly as the reference does - the binary32 screen runs on the
//           rounded ray, the spheres that pass get the binary64 Sphere.hit - a miss adds the binary64 sky
//           colour to its pixel and ends, a hit is shaded (binary32 from the rounded hit point on) and the
//           scattered path goes into the per-wave pool in LDS.  Lanes with a path in flight just wait.
//  BOUNCE   otherwise: idle lanes have taken paths from the pool; every lane with a path traces its next
//           segment in binary32 (scan, shade, scatter or end).
//
// K_LDS: -k per sphere in LDS (tables up to kMaxSpheresKInLds entries).  X_LDS: the records of the
// exact test (16 bytes per sphere) in LDS as well (tables up to kMaxSpheresExactInLds entries): a
// round of exact tests then waits for LDS, not for L2, and the survivor list has 16-bit entries.
// SLAB_AXIS: what camera rays see before the filter: the slab test along axis 0, 1 or 2, the interval test
// (kIntervalTest, see scan_interval), the mask test in front of it (kMaskTest, see scan_masked), the bundle test
// (kBundleTest, see scan_bundle), or nothing (-1).
template <bool K_LDS, bool X_LDS, int SLAB_AXIS>
__global__ void __launch_bounds__(kTraceThreads, 5) rtow_trace_kernel(const TraceParams P) {
  // LDS: [4 waves][WaveLds::kTotal], then (K_LDS) -k per sphere, then (X_LDS) the exact-test records
  extern __shared__ f4v lds_raw[];
  typedef WaveLds<X_LDS> L;
  uint32_t* const wave_lds = (uint32_t*)lds_raw + (threadIdx.x >> 6) * L::kTotal;
  // behind the waves' regions: -k of the whole table (K_LDS), or one ring of tiles of -k per wave (scan_tiled)
  // (the wave's number as a scalar: the LDS address of a global_load_lds must be wave-uniform, and the compiler cannot see
  // that threadIdx.x >> 6 is)
  const f4v* const nk_lds = lds_raw + (kTraceThreads / 64) * L::kTotal / 4 +
                            (K_LDS ? 0u : (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (uint32_t)kRingF4);
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t* const pool = wave_lds + L::kPool;
  const geom_ptr geom = (geom_ptr)(uintptr_t)P.geom;
  CandList<X_LDS> cand;
  cand.list = (typename CandList<X_LDS>::entry_t*)wave_lds;
  cand.best = (lds_u64_ptr)(unsigned long long*)(wave_lds + L::kKeys);
  cand.n = 0;
  cand.lane = lane;
  PrimaryCtx pc;
  pc.rays = (lds_f64_ptr)(const double*)pool;
  pc.best_id = (lds_u32_ptr)(wave_lds + L::kIds);
  SumCache sums;
  {
    uint32_t* const mine = wave_lds + L::kSums;
    if (lane < kSumCacheDwords) mine[lane] = 0u;   // read back only by this wave, in order
    sums.words = (lds_u64_ptr)(unsigned long long*)mine;
    for (uint32_t k = 0; k < kSumSlots; ++k) sums.tag[k] = kNoPixel;
    sums.victim = 0;
#ifdef RTOW_GUARD
    sums.n_pixels = (long long)P.n_rows * P.W;
#endif
  }
  if (K_LDS) {
    const int n4 = P.n_geom_pad / 4 + 2;
    f4v* const dst = lds_raw + (kTraceThreads / 64) * L::kTotal / 4;
    for (int i = threadIdx.x; i < n4; i += kTraceThreads) dst[i] = ((const f4v*)P.geom_nk)[i];
    if (X_LDS) {
      f4v* const xdst = dst + n4;
      for (int i = threadIdx.x; i < P.n_geom_pad; i += kTraceThreads) xdst[i] = ((const f4v*)P.geom_exact)[i];
    }
    __syncthreads();
  }
  typedef const f4v __attribute__((address_space(3)))* exact_lds_ptr;
  const exact_lds_ptr exact_lds = (exact_lds_ptr)(nk_lds + (P.n_geom_pad / 4 + 2));

  // wave-uniform: the chunk being handed out.  A chunk is a run of consecutive paths in
  // pixel-major order (all samples of a pixel, then the next pixel of the row list).
  uint32_t e_next = 0, e_end = 0;     // offsets inside the chunk
  unsigned long long seen = 0;        // first path of this wave's previous chunk
  uint32_t pool_n = 0, pool_head = 0; // scattered paths in the pool / already handed to lanes
  uint32_t p_base = 0, s_base = 0;    // pixel (index in the row list's image) and sample of the chunk's first path
  uint32_t r_base = 0, i_base = 0;    // row (in the row list) and column of that pixel
  bool exhausted = false;
  uint32_t nseg_primary = 0;          // primary segments traced by this wave

  // per lane: the path in flight
  bool alive = false;
  f3 o = {0, 0, 0}, d = {0, 0, 1};
  float tr = 0, tg = 0, tb = 0;
  Stream g = {0, 0};
  int depth = 0, prev = -1;
  uint32_t pix = 0;                   // its pixel (index in the row list's image)
  unsigned long long nseg_bounce = 0; // wave-uniform: later segments traced by this wave
  float my_paths = 0.0f;              // paths this wave has been handed so far
  const float inv_waves = 1.0f / (float)P.n_waves;
  RTOW_PROF_DECL;
  // the grid's first wave brackets its life - nearly the whole launch - with both clocks, so that the host can
  // tell what shader clock the chip sustained under this kernel
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    P.seg_counter[2] = __builtin_amdgcn_s_memtime();
    P.seg_counter[3] = __builtin_amdgcn_s_memrealtime();
  }
#ifdef RTOW_PROFILE_TIMELINE
  const uint32_t wave_id = blockIdx.x * (kTraceThreads / 64) + (threadIdx.x >> 6);
  if (lane == 0 && wave_id < 16384) g_wave_times[12 * wave_id] = __builtin_amdgcn_s_memrealtime();
  unsigned long long iters_ = 0;
#endif

  for (;;) {
    const unsigned long long t_refill = RTOW_CLOCK();
    // ---- hand pooled paths to the idle lanes
    {
      const uint64_t need = __ballot(!alive);
      const uint32_t avail = pool_n - pool_head;
      if (need != 0 && avail != 0) {
        const uint32_t rank = lanes_below(need);
        const uint32_t cnt = (uint32_t)__popcll(need);
        const bool take = !alive && rank < avail;
        const uint32_t entry = take ? pool_head + rank : 0u;
        if (take) {
          pix = pool[kPoolPix * 64 + entry];
          o.x = __uint_as_float(pool[(kPoolO + 0) * 64 + entry]); o.y = __uint_as_float(pool[(kPoolO + 1) * 64 + entry]);
          o.z = __uint_as_float(pool[(kPoolO + 2) * 64 + entry]); d.x = __uint_as_float(pool[(kPoolD + 0) * 64 + entry]);
          d.y = __uint_as_float(pool[(kPoolD + 1) * 64 + entry]); d.z = __uint_as_float(pool[(kPoolD + 2) * 64 + entry]);
          tr = __uint_as_float(pool[(kPoolT + 0) * 64 + entry]); tg = __uint_as_float(pool[(kPoolT + 1) * 64 + entry]);
          tb = __uint_as_float(pool[(kPoolT + 2) * 64 + entry]);
          g.x0 = pool[kPoolX0 * 64 + entry]; g.key1 = pool[kPoolKey1 * 64 + entry];
          prev = (int)pool[kPoolPrev * 64 + entry];
          depth = KARG(max_depth) - 1;
          alive = true;
        }
        RTOW_COUNT(23, min(cnt, avail));
        pool_head += min(cnt, avail);
      }
    }
    // ---- lanes still idle and nothing pooled: start the next paths of the chunk (a PRIMARY iteration)
    uint32_t n_gen = 0;
    if (__ballot(!alive) != 0 && pool_head >= pool_n) {
      if (e_next >= e_end && !exhausted) {
        // guided self-scheduling: a fixed fraction (1 / chunk_taper; a fraction of an even share) of what
        // was left when this wave last looked, clamped to [chunk_min, chunk_max] and rounded down to whole wavefronts
        uint32_t want = KARG(chunk_max);
        const unsigned long long left_seen = KARG(total_paths) - seen;
        if (left_seen < (unsigned long long)KARG(chunk_taper) * KARG(chunk_max)) {
          want = ((uint32_t)left_seen / KARG(chunk_taper)) & ~63u;
          want = want < KARG(chunk_min) ? KARG(chunk_min) : want;
        }
        unsigned long long first = 0;
        if (lane == 0) first = atomicAdd(KARG(path_counter), (unsigned long long)want);
        // (the builtin returns int: widen through uint32_t, or a low word >= 2^31 sign-extends over the high word)
        first = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(first >> 32)) << 32) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)first);
        if (first >= KARG(total_paths)) {
#ifdef RTOW_PROFILE_TIMELINE
          {
            const unsigned long long live_ = __popcll(__ballot(alive));
            if (lane == 0 && wave_id < 16384) {
              g_wave_times[12 * wave_id + 1] = __builtin_amdgcn_s_memrealtime();
              g_wave_times[12 * wave_id + 3] = live_;
              g_wave_times[12 * wave_id + 4] = iters_;
            }
          }
#endif
          exhausted = true;
        } else {
          seen = first;
          {
            // The SIMD's arbiter serves its resident waves by priority, then age, and age never
            // changes for persistent waves: left alone, the oldest wave of a SIMD runs ten times as
            // many iterations as the youngest, which then sits on stale work when the frame runs out
            // and keeps the grid alive for milliseconds.  Each wave therefore compares what it has
            // consumed with an even share of what has been handed out and raises its priority when it
            // is behind, lowers it when it is ahead.
            const float ahead = my_paths - (float)first * inv_waves;
            if (ahead < -512.0f) __builtin_amdgcn_s_setprio(3);
            else if (ahead < 0.0f) __builtin_amdgcn_s_setprio(2);
            else if (ahead < 512.0f) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            my_paths += (float)want;
          }
          const unsigned long long left = KARG(total_paths) - first;
          // Pixels are handed out from the LAST pixel of the row list backwards, so a frame ends on its
          // top rows.  The waves retire only when their longest path has ended, and in the scenes this
          // renderer is used on the top of the frame is sky (one segment per path) while the bottom is
          // ground and spheres (paths of up to max_depth segments): finishing on cheap paths cuts the
          // time the grid spends draining from milliseconds to one iteration.
          const uint32_t p_seq = (uint32_t)(first / (uint32_t)KARG(spp));
          s_base = (uint32_t)(first - (unsigned long long)p_seq * (uint32_t)KARG(spp));
          p_base = (uint32_t)KARG(n_rows) * (uint32_t)P.W - 1u - p_seq;
          r_base = p_base / (uint32_t)P.W;
          i_base = p_base - r_base * (uint32_t)P.W;
          e_next = 0;
          e_end = left < want ? (uint32_t)left : want;
        }
      }
      if (e_next < e_end) n_gen = min(64u, e_end - e_next);
    }
    const bool primary = n_gen != 0;
    if (!primary && __ballot(alive) == 0) break;
    const unsigned long long t_seg = RTOW_CLOCK();

#ifdef RTOW_PROFILE_COUNTERS
    int shade_kind = -2;   // -2 no segment, -1 sky, else the material kind
#endif
#if defined(RTOW_PROFILE_COUNTERS) || defined(RTOW_PROFILE_TIMELINE)
    unsigned long long t_scan0 = 0, t_scan1 = 0;
#define RTOW_STAMP(v) v = RTOW_CLOCK()
#else
#define RTOW_STAMP(v) do { } while (0)
#endif
    if (primary) {
      // ================= PRIMARY iteration: the chunk's next n_gen paths, one per lane =================
      // (pixel, sample) of lane's path: e counts samples from the first sample of the chunk's
      // first pixel.  With spp >= chunk_max it overflows spp at most once.
      const bool gen = lane < n_gen;
      const uint32_t e = s_base + e_next + (gen ? lane : 0u);
      uint32_t dp, sample;
      if ((uint32_t)KARG(spp) >= KARG(chunk_max)) {
        dp = e >= (uint32_t)KARG(spp) ? 1u : 0u;
        sample = e - (dp ? (uint32_t)KARG(spp) : 0u);
      } else {
        // e < 2 * chunk_max + spp and spp < chunk_max here: exact (2^32 / 1 does not fit inv_spp)
        dp = KARG(spp) == 1 ? e : __umulhi(e, KARG(inv_spp));
        sample = e - dp * (uint32_t)KARG(spp);
      }
      double* const rays = (double*)pool;   // the pool is empty during this iteration: its memory holds the rays
      {
        // The camera (19 doubles) is read from the kernel arguments here, every time: kept in scalar
        // registers across the sphere scan it would push other values out into vector-register lanes.
        typedef const double __attribute__((address_space(4)))* karg_ptr;
        karg_ptr cf = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();   // TraceParams starts with cam: 19 doubles
        asm volatile("" : "+s"(cf));
        static_assert(offsetof(TraceParams, cam) == 0 && sizeof(CameraParams64) == 19 * sizeof(double), "camera layout");
        uint32_t r = r_base;
        int ci = (int)i_base - (int)dp;
        while (ci < 0) { ci += P.W; --r; }
        const int y = KARG(rows) ? KARG(rows)[r] : (int)r;
        Stream g2 = stream_begin(KARG(seed_lo), KARG(seed_hi), (uint32_t)y * (uint32_t)P.W + (uint32_t)ci, sample + KARG(sample_base));
        // raytracer.nim:41-42: u before v, divisors W-1 and H-1, j counted from the bottom row
        const double xi1 = (double)(draw_u32(g2) >> 8) * 0x1p-24, xi2 = (double)(draw_u32(g2) >> 8) * 0x1p-24;
        const double cs = ((double)ci + xi1) / (double)(P.W - 1);
        const double ct = ((double)(P.H - 1 - y) + xi2) / (double)(P.H - 1);
        // randomInUnitDisk (basetypes.nim:130-134): x, y = rand(-1.0..1.0) = (k - 2^23) / 2^23 for a 24-bit draw k;
        // x^2 + y^2 >= 1 is decided exactly in integers (binary64 evaluates the same sum without rounding)
        int ax, ay;
        do {
          ax = (int)(draw_u32(g2) >> 8) - (1 << 23); ay = (int)(draw_u32(g2) >> 8) - (1 << 23);
        } while ((long long)ax * ax + (long long)ay * ay >= (1ll << 46));
        // getRay (camera.nim:34-39)
        const double lens = cf[18];
        const double rdx = lens * ((double)ax * 0x1p-23), rdy = lens * ((double)ay * 0x1p-23);
        double o64[3], d64[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const double off = cf[12 + k] * rdx + cf[15 + k] * rdy;
          o64[k] = cf[0 + k] + off;
          d64[k] = cf[3 + k] + cs * cf[6 + k] + ct * cf[9 + k] - cf[0 + k] - off;
        }
        // the |radius| >= 100 spheres never pass the screen (k = +inf): every lane tests them itself, in list
        // order with the shrinking t_max (hittables.nim:84-88)
        const double a64 = d64[0] * d64[0] + d64[1] * d64[1] + d64[2] * d64[2];   // length_squared (basetypes.nim:88)
        double closest = INFINITY;
        int pid = -1;
        for (int k = 0; k < KARG(n_big); ++k) {
          // wave-uniform records, read with scalar loads (constant address space: no kernel writes these tables)
          typedef const int32_t __attribute__((address_space(4)))* big_i32_ptr;
          typedef const double __attribute__((address_space(4)))* geom_f64_ptr;
          const int j = ((big_i32_ptr)(uintptr_t)KARG(big))[12 * k + 10];   // BigSphere::index
          const geom_f64_ptr g64 = (geom_f64_ptr)(uintptr_t)KARG(geom64) + 4 * (size_t)(RTOW_GUARDED(2, j, KARG(n_list)) ? j : 0);
          const double4 sp64 = make_double4(g64[0], g64[1], g64[2], g64[3]);
          double root;
          if (RTOW_GUARDED(2, j, KARG(n_list)) &&
              sphere_hit_f64(o64[0], o64[1], o64[2], d64[0], d64[1], d64[2], a64, sp64, root) && !(closest < root)) {
            closest = root;
            pid = j;
          }
        }
        rays[0 * 64 + lane] = o64[0]; rays[1 * 64 + lane] = o64[1]; rays[2 * 64 + lane] = o64[2];
        rays[3 * 64 + lane] = d64[0]; rays[4 * 64 + lane] = d64[1]; rays[5 * 64 + lane] = d64[2];
        pool[12 * 64 + lane] = g2.x0; pool[13 * 64 + lane] = g2.key1;   // the stream goes on in the hit's shading
        cand.best[lane] = (unsigned long long)__double_as_longlong(closest);
        pc.best_id[lane] = (uint32_t)(pid + 1);
        const f3 so = {(float)o64[0], (float)o64[1], (float)o64[2]};
        const f3 sd = {(float)d64[0], (float)d64[1], (float)d64[2]};
        f3 su;
        float tmin_unused;
        segment_begin(sd, su, tmin_unused);
        RTOW_COUNT(14, 1);
        RTOW_STAMP(t_scan0);
        // the screen runs on the binary32 roundings of the ray; survivors get the binary64 test
        // slab_inv_rmin sits behind the camera in the kernel arguments
        typedef const float __attribute__((address_space(4)))* karg_f32_ptr;
        static_assert(offsetof(TraceParams, slab_inv_rmin) == sizeof(CameraParams64), "slab_inv_rmin follows the camera");
        if constexpr (SLAB_AXIS == kMaskTest) {
          if (X_LDS) scan_masked<K_LDS>(geom, exact_lds, nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, (float)cs, (float)ct, pc RTOW_PROF_ARG);
          else scan_masked<K_LDS>(geom, (const f4v*)KARG(geom_exact), nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, (float)cs, (float)ct, pc RTOW_PROF_ARG);
        } else if constexpr (SLAB_AXIS == kIntervalTest) {
          if (X_LDS) scan_interval<K_LDS>(geom, exact_lds, nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, (float)cs, (float)ct, pc RTOW_PROF_ARG);
          else scan_interval<K_LDS>(geom, (const f4v*)KARG(geom_exact), nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, (float)cs, (float)ct, pc RTOW_PROF_ARG);
        } else if constexpr (SLAB_AXIS == kBundleTest) {
          // axis of the bundle: from the middle of the lens to the point the wave's first ray aims at (any axis is
          // valid - the spread of the rays is measured about it - this one sits in the middle of the lens)
          const f3 ax = {(float)cf[0], (float)cf[1], (float)cf[2]};
          const f3 aim = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, so.x + sd.x))) - ax.x,
                          __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, so.y + sd.y))) - ax.y,
                          __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, so.z + sd.z))) - ax.z};
          f3 aw;
          segment_begin(aim, aw, tmin_unused);
          if (X_LDS) scan_bundle<K_LDS>(geom, exact_lds, nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, ax, aw, pc RTOW_PROF_ARG);
          else scan_bundle<K_LDS>(geom, (const f4v*)KARG(geom_exact), nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, ax, aw, pc RTOW_PROF_ARG);
        } else {
          SlabRay sr = {f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}};
          if (SLAB_AXIS >= 0) sr = slab_operands<SLAB_AXIS < 0 ? 0 : SLAB_AXIS>(so, su, KARG(cmax), ((karg_f32_ptr)cf)[2 * 19]);
          if (X_LDS) scan_regular<true, K_LDS, SLAB_AXIS>(geom, exact_lds, nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, 0.0f, -1, pc, sr RTOW_PROF_ARG);
          else scan_regular<true, K_LDS, SLAB_AXIS>(geom, (const f4v*)KARG(geom_exact), nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), gen, so, su, 0.0f, -1, pc, sr RTOW_PROF_ARG);
        }
        RTOW_STAMP(t_scan1);
      }
      // ---- the camera rays' hits: sky ends the path (binary64, raytracer.nim:18-20), a hit scatters
      const double t64 = __longlong_as_double((long long)cand.best[lane]);
      const int pid = (int)pc.best_id[lane] - 1;
      const lds_f64_ptr ray = pc.rays + lane;
      const double ox = ray[0], oy = ray[64], oz = ray[128], dx = ray[192], dy = ray[256], dz = ray[320];
      Stream g2 = {pool[12 * 64 + lane], pool[13 * 64 + lane]};
      lds_order();   // every lane has read its ray before the pool entries below overwrite the rays
      const uint32_t pix2 = p_base - dp;
      uint64_t fr = 0, fg = 0, fb = 0;
      bool sky = false, cont = false;
      f3 p = {0, 0, 0}, dn = {0, 0, 1};
      float ptr_ = 1.0f, ptg_ = 1.0f, ptb_ = 1.0f;
      const bool struck = gen && pid >= 0 && RTOW_GUARDED(4, pid, KARG(n_list));
      f3 ball;
      float ball_l2;
      bool wants_ball = false;   // Lambertian and Metal scatter through randomInUnitSphere (hitrecord.nim:84,97)
      if (struck) wants_ball = KARG(shade)[pid].kind != kDielectric;
      unit_ball_wave(wants_ball, g2, ball, ball_l2, lane, (uint2*)cand.list RTOW_PROF_ARG);
      if (gen) {
        if (KARG(hit_id) && sample + KARG(sample_base) == 0u && RTOW_GUARDED(3, pix2, (long long)KARG(n_rows) * P.W)) KARG(hit_id)[pix2] = pid;
        if (!struck) {
          // unitVector(r.dir).y = dir.y / length(dir); t = 0.5 (y + 1); (1 - t) * 1 + t * colour
          const double uy = dy / sqrt(dx * dx + dy * dy + dz * dz);
          const double ts = 0.5 * (uy + 1.0);
          const double w1 = (1.0 - ts) * 1.0;
          fr = to_fix64(w1 + ts * 0.5); fg = to_fix64(w1 + ts * 0.7); fb = to_fix64(w1 + ts * 1.0);
          sky = true;
        } else {
          // rec.p = r.at(t) = orig + t * dir (basetypes.nim:170), rounded to binary32: the path goes on from
          // there, with the binary32 unit direction of the rounded camera ray
          p = {(float)(ox + t64 * dx), (float)(oy + t64 * dy), (float)(oz + t64 * dz)};
          const f3 sd = {(float)dx, (float)dy, (float)dz};
          f3 su;
          float tmin_unused;
          segment_begin(sd, su, tmin_unused);
          const SphereShade sh = KARG(shade)[pid];
          cont = shade_at_ball(sh, p, su, g2, ball, ball_l2, dn, ptr_, ptg_, ptb_) && KARG(max_depth) > 1;
        }
      }
      RTOW_COUNT(21, __popcll(__ballot(sky)));
      RTOW_COUNT(22, __popcll(__ballot(gen && !sky)));
      flush_sums(sums, sky, pix2, fr, fg, fb, lane);
      const uint64_t m = __ballot(cont);
      if (cont) {
        const uint32_t slot = lanes_below(m);
        pool[(kPoolO + 0) * 64 + slot] = __float_as_uint(p.x); pool[(kPoolO + 1) * 64 + slot] = __float_as_uint(p.y);
        pool[(kPoolO + 2) * 64 + slot] = __float_as_uint(p.z); pool[(kPoolD + 0) * 64 + slot] = __float_as_uint(dn.x);
        pool[(kPoolD + 1) * 64 + slot] = __float_as_uint(dn.y); pool[(kPoolD + 2) * 64 + slot] = __float_as_uint(dn.z);
        pool[(kPoolT + 0) * 64 + slot] = __float_as_uint(ptr_); pool[(kPoolT + 1) * 64 + slot] = __float_as_uint(ptg_);
        pool[(kPoolT + 2) * 64 + slot] = __float_as_uint(ptb_);
        pool[kPoolX0 * 64 + slot] = g2.x0; pool[kPoolKey1 * 64 + slot] = g2.key1;
        pool[kPoolPix * 64 + slot] = pix2; pool[kPoolPrev * 64 + slot] = (uint32_t)pid;
      }
      pool_n = (uint32_t)__popcll(m);
      pool_head = 0;
      e_next += n_gen;
      nseg_primary += n_gen;
      lds_order();   // the entries are read by other lanes of this wave
    } else {
      // ================= BOUNCE iteration: every path in flight traces its next segment in binary32 =================
      f3 u = {0, 0, 1};
      float tmin = 0.0f;
      {
        float best = INFINITY;
        int id = -1;
        if (alive) {
          segment_begin(d, u, tmin);
          // (read through the constant address space: wave-uniform scalar loads into SGPRs - the table is written by
          // no kernel - instead of one vector load per lane from the same address)
          typedef const double __attribute__((address_space(4)))* big_f64_ptr;
          typedef const int32_t __attribute__((address_space(4)))* big_i32_ptr;
          static_assert(sizeof(BigSphere) == 48 && offsetof(BigSphere, rr) == 32 && offsetof(BigSphere, index) == 40, "BigSphere layout");
          const big_f64_ptr big = (big_f64_ptr)(uintptr_t)KARG(big);
          for (int k = 0; k < KARG(n_big); ++k) {
            BigSphere bs;
            bs.cx = big[6 * k]; bs.cy = big[6 * k + 1]; bs.cz = big[6 * k + 2]; bs.r = big[6 * k + 3]; bs.rr = big[6 * k + 4];
            bs.index = ((big_i32_ptr)(big + 6 * k + 5))[0]; bs.pad_ = 0;
            float t;
            const bool ok = big_candidate(o, u, bs, tmin, bs.index == prev, t);
            offer(ok, t, bs.index, best, id);
          }
        }
        cand.best[lane] = hit_key(best, id);   // what the binary64 spheres left; read back after the last drain
      }
      nseg_bounce += (unsigned long long)__popcll(__ballot(alive));
      RTOW_COUNT(4, __popcll(__ballot(alive)));
      RTOW_STAMP(t_scan0);
      // all 64 lanes take part: lanes without a path offer no candidates but help with the exact tests
      if (X_LDS && exhausted && (uint32_t)__popcll(__ballot(alive)) <= kSparseLanes)
        scan_sparse(exact_lds, nk_lds, cand, KARG(n_geom_pad), KARG(cmax), alive, o, u, tmin, prev, pc RTOW_PROF_ARG);
      else if (X_LDS) scan_regular<false, K_LDS, -1>(geom, exact_lds, nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), alive, o, u, tmin, prev, pc, SlabRay{f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}} RTOW_PROF_ARG);
      else scan_regular<false, K_LDS, -1>(geom, (const f4v*)KARG(geom_exact), nk_lds, cand, KARG(n_geom_pad) / 4, KARG(cmax), alive, o, u, tmin, prev, pc, SlabRay{f2v{0.0f, 0.0f}, f2v{0.0f, 0.0f}} RTOW_PROF_ARG);
      RTOW_STAMP(t_scan1);
      // A path that ends adds its radiance to its pixel right away, through the wave's cache of pixel sums
      // (paths that end absorbed or at the depth limit are black and add nothing).
      uint32_t fr = 0, fg = 0, fb = 0;
      // world.hit + scatter (raytracer.nim:11-16)
      const unsigned long long key = cand.best[lane];
      const float best = __uint_as_float((uint32_t)(key >> 32));
      const int id = hit_key_id(key);
      const bool struck = alive && id >= 0 && RTOW_GUARDED(5, id, KARG(n_list));
      f3 ball;
      float ball_l2;
      bool wants_ball = false;   // Lambertian and Metal scatter through randomInUnitSphere (hitrecord.nim:84,97)
      if (struck) wants_ball = KARG(shade)[id].kind != kDielectric;
      unit_ball_wave(wants_ball, g, ball, ball_l2, lane, (uint2*)cand.list RTOW_PROF_ARG);
      if (alive) {
        float rr_ = 0.0f, rg_ = 0.0f, rb_ = 0.0f;
        bool ended;
        if (!struck) {
          shade_sky(u, tr, tg, tb, rr_, rg_, rb_);
          ended = true;
#ifdef RTOW_PROFILE_COUNTERS
          shade_kind = -1;
#endif
        } else {
          const SphereShade sh = KARG(shade)[id];
          f3 dn;
          const f3 p = {fmaf(best, u.x, o.x), fmaf(best, u.y, o.y), fmaf(best, u.z, o.z)};
          o = p;   // the scattered ray starts at the hit point whether or not it survives
          const bool scattered = shade_at_ball(sh, p, u, g, ball, ball_l2, dn, tr, tg, tb);
#ifdef RTOW_PROFILE_COUNTERS
          shade_kind = sh.kind;
#endif
          d = dn;
          prev = id;
          --depth;
          ended = !scattered || depth <= 0;  // absorbed, or depth exhausted: black (raytracer.nim:8-9,16)
        }
        if (ended) {
          fr = to_fix(rr_); fg = to_fix(rg_); fb = to_fix(rb_);
          alive = false;
        }
      }
      flush_sums(sums, (fr | fg | fb) != 0u, pix, widen_fix(fr), widen_fix(fg), widen_fix(fb), lane);
    }
    RTOW_COUNT(3, 1);
#ifdef RTOW_PROFILE_COUNTERS
    {
      RTOW_COUNT(16, __popcll(__ballot(shade_kind == -1)));
      RTOW_COUNT(17, __popcll(__ballot(shade_kind == kLambertian)));
      RTOW_COUNT(18, __popcll(__ballot(shade_kind == kMetal)));
      RTOW_COUNT(19, __popcll(__ballot(shade_kind == kDielectric)));
    }
#endif
#ifdef RTOW_PROFILE_TIMELINE
    ++iters_;
    if (lane == 0 && wave_id < 16384 && exhausted) {
      g_wave_times[12 * wave_id + 6] = t_seg; g_wave_times[12 * wave_id + 7] = t_scan0;
      g_wave_times[12 * wave_id + 8] = t_scan1; g_wave_times[12 * wave_id + 9] = RTOW_CLOCK();
      g_wave_times[12 * wave_id + 11] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#ifdef RTOW_PROFILE_COUNTERS
    RTOW_COUNT(5, t_seg - t_refill);                    // hand-out, chunk fetch (both kinds of iteration)
    RTOW_COUNT(primary ? 1 : 6, t_scan0 - t_seg);       // ray set-up: camera rays + binary64 spheres / normalise + binary64 spheres
    RTOW_COUNT(primary ? 2 : 7, t_scan1 - t_scan0);     // scan (filter + exact tests)
    RTOW_COUNT(primary ? 13 : 8, RTOW_CLOCK() - t_scan1);  // shading (+ pool entries)
    RTOW_COUNT(15, RTOW_CLOCK() - t_refill);            // whole iteration
#endif
    (void)t_refill; (void)t_seg;
  }

#ifdef RTOW_PROFILE_TIMELINE
  if (lane == 0 && wave_id < 16384) g_wave_times[12 * wave_id + 10] = __builtin_amdgcn_s_memrealtime();
#endif
  sum_cache_drain(sums, lane);
  RTOW_PROF_FLUSH();
#ifdef RTOW_PROFILE_TIMELINE
  if (lane == 0 && wave_id < 16384) {
    g_wave_times[12 * wave_id + 2] = __builtin_amdgcn_s_memrealtime();
    g_wave_times[12 * wave_id + 5] = iters_;
  }
#endif
  // ray-segment count: wave reduction, one atomic per wave
  const unsigned long long total = nseg_bounce + nseg_primary;
  if (lane == 0) atomicAdd(KARG(seg_counter), total);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    KARG(seg_counter)[4] = __builtin_amdgcn_s_memtime();
    KARG(seg_counter)[5] = __builtin_amdgcn_s_memrealtime();
  }
}

__global__ void __launch_bounds__(256) rtow_resolve_kernel(const unsigned long long* __restrict__ accum,
                                                           uint8_t* __restrict__ rgb, uint32_t n_channels,
                                                           int spp) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_channels) rgb[i] = (uint8_t)quantise_fix(accum[i], spp);
}

static size_t wave_lds_bytes(bool short_entries) {
  return sizeof(uint32_t) * (kTraceThreads / 64) * (short_entries ? WaveLds<true>::kTotal : WaveLds<false>::kTotal);
}
static size_t ring_lds_bytes() { return wave_lds_bytes(false) + sizeof(float4) * (size_t)kRingF4 * (kTraceThreads / 64); }   // scan_tiled's rings
static size_t k_lds_bytes(int n_geom_pad) { return wave_lds_bytes(false) + sizeof(float) * (size_t)(n_geom_pad + 8); }
static size_t kx_lds_bytes(int n_geom_pad) {
  return wave_lds_bytes(true) + sizeof(float) * (size_t)(n_geom_pad + 8) + sizeof(float4) * (size_t)n_geom_pad;
}

template <int SLAB_AXIS>
static void launch_trace_axis(const TraceParams& p, int grid_blocks, hipStream_t stream) {
  if (p.n_geom_pad <= kMaxSpheresExactInLds)
    hipLaunchKernelGGL((rtow_trace_kernel<true, true, SLAB_AXIS>), dim3(grid_blocks), dim3(kTraceThreads), kx_lds_bytes(p.n_geom_pad),
                       stream, p);
  else if (p.n_geom_pad <= kMaxSpheresKInLds)
    hipLaunchKernelGGL((rtow_trace_kernel<true, false, SLAB_AXIS>), dim3(grid_blocks), dim3(kTraceThreads), k_lds_bytes(p.n_geom_pad),
                       stream, p);
  else
    hipLaunchKernelGGL((rtow_trace_kernel<false, false, SLAB_AXIS>), dim3(grid_blocks), dim3(kTraceThreads), ring_lds_bytes(), stream, p);
}

void launch_trace(const TraceParams& p, int slab_axis, int grid_blocks, hipStream_t stream) {
  switch (slab_axis) {
    case 0: launch_trace_axis<0>(p, grid_blocks, stream); break;
    case 1: launch_trace_axis<1>(p, grid_blocks, stream); break;
    case 2: launch_trace_axis<2>(p, grid_blocks, stream); break;
    case kBundleTest: launch_trace_axis<kBundleTest>(p, grid_blocks, stream); break;
    case kIntervalTest: launch_trace_axis<kIntervalTest>(p, grid_blocks, stream); break;
    case kMaskTest: launch_trace_axis<kMaskTest>(p, grid_blocks, stream); break;
    default: launch_trace_axis<-1>(p, grid_blocks, stream); break;
  }
}

void launch_resolve(const unsigned long long* accum, uint8_t* rgb, uint32_t n_channels, int spp,
                    hipStream_t stream) {
  if (n_channels == 0) return;
  hipLaunchKernelGGL(rtow_resolve_kernel, dim3((n_channels + 255) / 256), dim3(256), 0, stream, accum, rgb,
                     n_channels, spp);
}

// ------------------------------------------------------------------ PPM text (raytracer.nim:28,35)
// writeColor prints three decimal integers and a newline per pixel.  At 4K that is ~100 MB of
// text, seconds of fprintf on the host; on the device it is a byte-shuffling pass bounded by HBM
// (3 bytes read, 6-12 written per pixel): lengths per workgroup, a prefix sum, then every
// workgroup composes its lines in LDS and stores them as one contiguous run.
static __device__ __forceinline__ uint32_t dec_digits(uint32_t v) { return v >= 100u ? 3u : (v >= 10u ? 2u : 1u); }

static __device__ __forceinline__ uint32_t pixel_text_len(const uint8_t* rgb, uint32_t pix, uint32_t n_pixels,
                                                          uint32_t& r, uint32_t& g, uint32_t& b) {
  if (pix >= n_pixels) { r = g = b = 0; return 0u; }
  r = rgb[3ull * pix]; g = rgb[3ull * pix + 1]; b = rgb[3ull * pix + 2];
  return dec_digits(r) + dec_digits(g) + dec_digits(b) + 3u;   // two blanks and the newline
}

// inclusive prefix sum over the kPpmBlock threads of a workgroup; returns this thread's inclusive
// value, `total` = sum over the workgroup
static __device__ __forceinline__ uint32_t block_inclusive_scan(uint32_t v, uint32_t* wave_tot, uint32_t& total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(v, off, 64);
    if ((int)lane >= off) v += up;
  }
  if (lane == 63u) wave_tot[wave] = v;
  __syncthreads();
  uint32_t before = 0;
  total = 0;
#pragma unroll
  for (uint32_t w = 0; w < kPpmBlock / 64; ++w) {
    const uint32_t t = wave_tot[w];
    before += w < wave ? t : 0u;
    total += t;
  }
  __syncthreads();
  return v + before;
}

__global__ void __launch_bounds__(kPpmBlock) ppm_lengths_kernel(const uint8_t* __restrict__ rgb, uint32_t n_pixels,
                                                               uint32_t* __restrict__ block_len) {
  __shared__ uint32_t wave_tot[kPpmBlock / 64];
  uint32_t r, g, b, total;
  const uint32_t len = pixel_text_len(rgb, blockIdx.x * kPpmBlock + threadIdx.x, n_pixels, r, g, b);
  (void)block_inclusive_scan(len, wave_tot, total);
  if (threadIdx.x == 0) block_len[blockIdx.x] = total;
}

// one workgroup of 1024 threads: thread t owns a contiguous span of the entries
__global__ void __launch_bounds__(1024) ppm_scan_kernel(const uint32_t* __restrict__ block_len, uint32_t n_blocks,
                                                         unsigned long long* __restrict__ block_off) {
  __shared__ unsigned long long span_sum[1024];
  const uint32_t per = (n_blocks + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per, hi = lo + per < n_blocks ? lo + per : n_blocks;
  unsigned long long s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += block_len[i];
  span_sum[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0;
    for (int i = 0; i < 1024; ++i) { const unsigned long long t = span_sum[i]; span_sum[i] = run; run += t; }
    block_off[n_blocks] = run;
  }
  __syncthreads();
  unsigned long long run = span_sum[threadIdx.x];
  for (uint32_t i = lo; i < hi; ++i) { block_off[i] = run; run += block_len[i]; }
}

static __device__ __forceinline__ uint32_t put_decimal(char* dst, uint32_t at, uint32_t v, char after) {
  if (v >= 100u) dst[at++] = (char)('0' + v / 100u);
  if (v >= 10u) dst[at++] = (char)('0' + (v / 10u) % 10u);
  dst[at++] = (char)('0' + v % 10u);
  dst[at++] = after;
  return at;
}

__global__ void __launch_bounds__(kPpmBlock) ppm_text_kernel(const uint8_t* __restrict__ rgb, uint32_t n_pixels,
                                                            const unsigned long long* __restrict__ block_off,
                                                            char* __restrict__ text) {
  __shared__ uint32_t wave_tot[kPpmBlock / 64];
  __shared__ char lines[kPpmBlock * 12];
  uint32_t r, g, b, total;
  const uint32_t len = pixel_text_len(rgb, blockIdx.x * kPpmBlock + threadIdx.x, n_pixels, r, g, b);
  uint32_t at = block_inclusive_scan(len, wave_tot, total) - len;
  if (len) {
    at = put_decimal(lines, at, r, ' ');
    at = put_decimal(lines, at, g, ' ');
    (void)put_decimal(lines, at, b, '\n');
  }
  __syncthreads();
  char* const dst = text + block_off[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < total; i += kPpmBlock) dst[i] = lines[i];
}

void launch_ppm_lengths(const uint8_t* rgb, uint32_t n_pixels, uint32_t* block_len, hipStream_t stream) {
  if (n_pixels == 0) return;
  hipLaunchKernelGGL(ppm_lengths_kernel, dim3((n_pixels + kPpmBlock - 1) / kPpmBlock), dim3(kPpmBlock), 0, stream, rgb,
                     n_pixels, block_len);
}

void launch_ppm_scan(const uint32_t* block_len, uint32_t n_blocks, unsigned long long* block_off, hipStream_t stream) {
  hipLaunchKernelGGL(ppm_scan_kernel, dim3(1), dim3(1024), 0, stream, block_len, n_blocks, block_off);
}

void launch_ppm_text(const uint8_t* rgb, uint32_t n_pixels, const unsigned long long* block_off, char* text,
                     hipStream_t stream) {
  if (n_pixels == 0) return;
  hipLaunchKernelGGL(ppm_text_kernel, dim3((n_pixels + kPpmBlock - 1) / kPpmBlock), dim3(kPpmBlock), 0, stream, rgb,
                     n_pixels, block_off, text);
}

// ------------------------------------------------------------------ self test of rt_sqrt / rt_rcp
// every binary32 bit pattern against the compiler's correctly rounded sqrtf(x) and 1.0f / x
__global__ void __launch_bounds__(256) arithmetic_selftest_kernel(unsigned long long* mismatches) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned int bad_sqrt = 0, bad_rcp = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    const float a = rt_sqrt(x), b = sqrtf(x), c = rt_rcp(x), d = 1.0f / x;
    bad_sqrt += !(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b));
    bad_rcp += !(__float_as_uint(c) == __float_as_uint(d) || (c != c && d != d));
  }
  if (bad_sqrt) atomicAdd(&mismatches[0], (unsigned long long)bad_sqrt);
  if (bad_rcp) atomicAdd(&mismatches[1], (unsigned long long)bad_rcp);
}

void launch_arithmetic_selftest(unsigned long long* mismatches, hipStream_t stream) {
  hipLaunchKernelGGL(arithmetic_selftest_kernel, dim3(256 * 8), dim3(256), 0, stream, mismatches);
}

// ------------------------------------------------------------------ single rays (rayColor / hit / getRay)
// One lane per ray, no scheduling: the reference's per-ray procs for callers that probe the world ray by
// ray.  The arithmetic is the renderer's: the first segment in binary64 with the reference's operations
// (hitrecord.nim:55-71 over hittables.nim:79-88, here for a caller-given [t_min, t_max]), later segments in
// binary32 with the same screen, exact test and shading functions the trace kernel uses, so a ray traced
// here gets what the same ray would get as a camera ray inside a render.
static __device__ __forceinline__ bool sphere_hit_range_f64(double ox, double oy, double oz, double dx, double dy, double dz,
                                                            double a, double4 sp, double t_min, double t_max, double& root) {
  const double ocx = ox - sp.x, ocy = oy - sp.y, ocz = oz - sp.z;
  double half_b = 0.0;
  half_b += ocx * dx; half_b += ocy * dy; half_b += ocz * dz;
  const double c = (ocx * ocx + ocy * ocy + ocz * ocz) - sp.w * sp.w;
  const double disc = half_b * half_b - a * c;
  if (disc < 0) return false;
  const double sqrtd = sqrt(disc);
  root = (-half_b - sqrtd) / a;
  if (root < t_min || t_max < root) {
    root = (-half_b + sqrtd) / a;
    if (root < t_min || t_max < root) return false;
  }
  return true;
}

// closest hit of a binary32 segment, one lane walking the whole list: the screen, then the exact test, then
// the closest-so-far rule - the arithmetic of scan_regular / cand_drain without the sharing across lanes
static __device__ __forceinline__ int closest_hit_serial(const RayParams& P, f3 o, f3 u, float tmin, int prev, float& best) {
  best = INFINITY;
  int id = -1;
  for (int k = 0; k < P.n_big; ++k) {
    const BigSphere bs = P.big[k];
    float t;
    const bool ok = big_candidate(o, u, bs, tmin, bs.index == prev, t);
    offer(ok, t, bs.index, best, id);
  }
  const FilterRay fr = filter_ray(o, u, P.cmax);
  for (int j = 0; j < P.n_list; ++j) {
    const float nk = P.geom_nk[j];          // -inf for the spheres handled above: they never pass
    const float4 sp = P.geom_exact[j];      // (cx, cy, cz, r^2)
    if (!(filter_value(sp.x, sp.y, sp.z, -nk, u, fr) >= fr.threshold)) continue;
    const float ocx = o.x - sp.x, ocy = o.y - sp.y, ocz = o.z - sp.z;
    float b = ocx * u.x; b = fmaf(ocy, u.y, b); b = fmaf(ocz, u.z, b);
    float m = fmaf(-ocx, ocx, sp.w); m = fmaf(-ocy, ocy, m); m = fmaf(-ocz, ocz, m);
    const float disc = fmaf(b, b, m);
    float t;
    const bool ok = regular_candidate(b, disc, tmin, j == prev, t);
    offer(disc >= 0.0f && ok, t, j, best, id);
  }
  return id;
}

__global__ void __launch_bounds__(256) rtow_rays_kernel(const RayParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const double ox = P.orig[3 * i], oy = P.orig[3 * i + 1], oz = P.orig[3 * i + 2];
  const double dx = P.dir[3 * i], dy = P.dir[3 * i + 1], dz = P.dir[3 * i + 2];
  const double a = dx * dx + dy * dy + dz * dz;
  // ---- hit(world, r, t_min, t_max, rec) in binary64, list order, shrinking t_max
  f3 o = {(float)ox, (float)oy, (float)oz};
  // The screen and the shading use the binary32 unit direction.  The caller's direction is binary64 and not
  // normalised: one whose length leaves binary32's range (below about 1e-19 or above 1e19, where |d|^2 under- or
  // overflows) is first scaled by a power of two - exact in binary64, and the unit direction of a ray does not
  // change by it, bit for bit - so that its largest component lands in [1, 2).
  double sdx = dx, sdy = dy, sdz = dz;
  {
    const double dm = fmax(fabs(dx), fmax(fabs(dy), fabs(dz)));
    if ((dm < 0x1p-60 || dm > 0x1p60) && dm > 0.0 && dm < INFINITY) {
      const int e = -ilogb(dm);
      sdx = ldexp(dx, e); sdy = ldexp(dy, e); sdz = ldexp(dz, e);
    }
  }
  f3 d = {(float)sdx, (float)sdy, (float)sdz};
  f3 u;
  float tmin;
  segment_begin(d, u, tmin);
  const FilterRay fr = filter_ray(o, u, P.cmax);
  double closest = P.t_max;
  int id = -1;
  for (int j = 0; j < P.n_list; ++j) {
    const float nk = P.geom_nk[j];
    if (nk != -INFINITY) {   // a binary32-table sphere: the render's screen runs first (it only drops misses)
      const float4 sp = P.geom_exact[j];
      if (!(filter_value(sp.x, sp.y, sp.z, -nk, u, fr) >= fr.threshold)) continue;
    }
    double root;
    if (sphere_hit_range_f64(ox, oy, oz, dx, dy, dz, a, P.geom64[j], P.t_min, closest, root)) { closest = root; id = j; }
  }
  if (P.hit_id) P.hit_id[i] = id;
  if (P.t) P.t[i] = id >= 0 ? closest : INFINITY;
  if (!P.color || P.max_depth <= 0) {
    if (P.color) { P.color[3 * i] = 0.0; P.color[3 * i + 1] = 0.0; P.color[3 * i + 2] = 0.0; }   // depth <= 0 (raytracer.nim:8-9)
    return;
  }
  // ---- rayColor(r, world, depth) (raytracer.nim:5-20); the stream is that of path (pixel i, sample 0)
  double cr = 0.0, cg = 0.0, cb = 0.0;
  if (id < 0) {
    const double uy = dy / sqrt(a);
    const double ts = 0.5 * (uy + 1.0);
    const double w1 = (1.0 - ts) * 1.0;
    cr = w1 + ts * 0.5; cg = w1 + ts * 0.7; cb = w1 + ts * 1.0;
  } else {
    Stream g = stream_begin(P.seed_lo, P.seed_hi, (uint32_t)i, 0u);
    float tr = 1.0f, tg = 1.0f, tb = 1.0f;
    f3 p = {(float)(ox + closest * dx), (float)(oy + closest * dy), (float)(oz + closest * dz)};
    f3 dn;
    bool alive = shade_at(P.shade[id], p, u, g, dn, tr, tg, tb);
    int prev = id;
    o = p;
    for (int depth = P.max_depth - 1; alive && depth > 0; --depth) {
      segment_begin(dn, u, tmin);
      float best;
      const int h = closest_hit_serial(P, o, u, tmin, prev, best);
      if (h < 0) {
        float rr, rg, rb;
        shade_sky(u, tr, tg, tb, rr, rg, rb);
        cr = rr; cg = rg; cb = rb;
        break;
      }
      alive = shade_hit(P.shade[h], o, u, best, g, dn, tr, tg, tb);
      prev = h;
    }
  }
  P.color[3 * i] = cr; P.color[3 * i + 1] = cg; P.color[3 * i + 2] = cb;
}

// getRay(camera, s, t) (camera.nim:34-39) for n (s, t) pairs in binary64; ray i draws its lens sample
// (basetypes.nim:130-134) from the stream of path (pixel i, sample 0), starting at that stream's first draw
__global__ void __launch_bounds__(256) rtow_get_rays_kernel(const RayParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  Stream g = stream_begin(P.seed_lo, P.seed_hi, (uint32_t)i, 0u);
  int ax, ay;
  do {
    ax = (int)(draw_u32(g) >> 8) - (1 << 23); ay = (int)(draw_u32(g) >> 8) - (1 << 23);
  } while ((long long)ax * ax + (long long)ay * ay >= (1ll << 46));
  const double lens = P.cam.lens_radius;
  const double rdx = lens * ((double)ax * 0x1p-23), rdy = lens * ((double)ay * 0x1p-23);
  const double cs = P.orig[i], ct = P.dir[i];   // the s and t arrays travel in the two input slots
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double off = P.cam.u[k] * rdx + P.cam.v[k] * rdy;
    P.color[3 * i + k] = P.cam.origin[k] + off;
    P.t[3 * i + k] = P.cam.llc[k] + cs * P.cam.hor[k] + ct * P.cam.ver[k] - P.cam.origin[k] - off;
  }
}

// scatter*(m, r_in, rec, attenuation, scattered) (hitrecord.nim:127-132) for n independent triples, in the
// arithmetic the renderer scatters with: binary32 on the rounded inputs, the incoming direction normalised
// first (Metal, Dielectric and the sky use the unit direction, hitrecord.nim:96,112).  Triple i draws from
// the stream of path (pixel i, sample 0), starting at that stream's first draw.
__global__ void __launch_bounds__(256) rtow_scatter_kernel(const ScatterParams P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  const f3 d = {(float)P.dir[3 * i], (float)P.dir[3 * i + 1], (float)P.dir[3 * i + 2]};
  const f3 n = {(float)P.normal[3 * i], (float)P.normal[3 * i + 1], (float)P.normal[3 * i + 2]};
  f3 u;
  float tmin_unused;
  segment_begin(d, u, tmin_unused);
  SphereShade s;
  s.cx = s.cy = s.cz = 0.0f; s.inv_r = 1.0f; s.pad_ = 0.0f;
  s.kind = (int32_t)P.kind[i];
  s.fuzz = (float)P.fuzz[i];
  s.ir = (float)P.ir[i];
  s.inv_ir = 1.0f / s.ir;
  if (s.kind == kDielectric) {   // a dielectric's record carries its two Schlick constants, attenuation is white
    s.ar = schlick_r0(s.inv_ir); s.ag = schlick_r0(s.ir); s.ab = 1.0f;
  } else {
    s.ar = (float)P.albedo[3 * i]; s.ag = (float)P.albedo[3 * i + 1]; s.ab = (float)P.albedo[3 * i + 2];
  }
  Stream g = stream_begin(P.seed_lo, P.seed_hi, (uint32_t)i, 0u);
  f3 b = {0.0f, 0.0f, 0.0f};
  float l2 = 0.0f;
  if (s.kind != kDielectric) b = unit_ball(g, l2);
  float tr = 1.0f, tg = 1.0f, tb = 1.0f;
  f3 nd = {0.0f, 0.0f, 0.0f};
  const bool ok = scatter_ball<true>(s, n, P.front[i] != 0, u, g, b, l2, nd, tr, tg, tb);
  P.attenuation[3 * i] = tr; P.attenuation[3 * i + 1] = tg; P.attenuation[3 * i + 2] = tb;
  P.out_dir[3 * i] = nd.x; P.out_dir[3 * i + 1] = nd.y; P.out_dir[3 * i + 2] = nd.z;
  P.ok[i] = ok ? 1 : 0;
}

void launch_scatter(const ScatterParams& p, hipStream_t stream) {
  if (p.n <= 0) return;
  hipLaunchKernelGGL(rtow_scatter_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p);
}

void launch_rays(const RayParams& p, bool get_rays, hipStream_t stream) {
  if (p.n <= 0) return;
  if (get_rays) hipLaunchKernelGGL(rtow_get_rays_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL(rtow_rays_kernel, dim3((p.n + 255) / 256), dim3(256), 0, stream, p);
}

#ifdef RTOW_GUARD
extern "C" int rtow_debug_guard(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_guard), sizeof(g_guard)) != hipSuccess) return -2;
  return 0;
}
#endif

#ifdef RTOW_PROFILE_COUNTERS
extern "C" int rtow_debug_counters(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(g_prof)) != hipSuccess) return -2;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return -2; }
  return 0;
}
#endif

#ifdef RTOW_PROFILE_TIMELINE
extern "C" int rtow_debug_wave_times(unsigned long long* out, int n_waves) {
  if (n_waves > 16384) n_waves = 16384;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), sizeof(unsigned long long) * 12 * n_waves) != hipSuccess) return -2;
  return 0;
}
#endif

int trace_blocks_per_cu(int n_geom_pad) {
  int nb = 0;
  hipError_t e;
  if (n_geom_pad <= kMaxSpheresExactInLds)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<true, true, 1>, kTraceThreads, kx_lds_bytes(n_geom_pad));
  else if (n_geom_pad <= kMaxSpheresKInLds)
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<true, false, 1>, kTraceThreads, k_lds_bytes(n_geom_pad));
  else
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rtow_trace_kernel<false, false, 1>, kTraceThreads, ring_lds_bytes());
  if (e != hipSuccess || nb < 1) nb = 1;
  return nb;
}

}  // namespace rtow
