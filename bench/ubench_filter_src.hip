// Where should the sphere table of the packed filter come from?  The renderer's scan reads the
// centres of a group of four spheres through the scalar cache into SGPRs (one s_load_dwordx16) and
// -k through a broadcast ds_read_b128 into VGPRs.  A packed FMA with an SGPR-pair source issues a
// little slower than one with VGPR sources only, so this probe runs the same loop body - 14 packed
// FMAs, the 4-way max and the compare per group, two groups per trip, operands requested one group
// ahead - with NLDS = 1..4 of the four quads (-k, cx, cy, cz) coming from LDS and the rest from
// SGPRs, at the renderer's occupancy (5 waves per SIMD, 256-thread blocks), and reports SIMD cycles
// per group and wave.  (Rays aim at the middle of the lattice, so about a quarter of the groups has a
// survivor in some lane - three times the renderer's rate; the cycles include those pushes.)
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o ubench_filter_src ubench_filter_src.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef const f16v __attribute__((address_space(4)))* geom_ptr;
typedef const f4v __attribute__((address_space(3)))* lds_ptr;

constexpr int kGroups = 122;   // 488 spheres, like the book scene's padded table
constexpr int kThreads = 256;

struct Ray { f2v ux, uy, uz, nou, ox2, oy2, oz2; float threshold; };

static __device__ __forceinline__ f2v pair_filter(f2v cx, f2v cy, f2v cz, f2v nk, const Ray& r) {
  f2v n = __builtin_elementwise_fma(cx, r.ux, r.nou);
  n = __builtin_elementwise_fma(cy, r.uy, n);
  n = __builtin_elementwise_fma(cz, r.uz, n);
  f2v t = __builtin_elementwise_fma(cx, r.ox2, nk);
  t = __builtin_elementwise_fma(cy, r.oy2, t);
  t = __builtin_elementwise_fma(cz, r.oz2, t);
  return __builtin_elementwise_fma(n, n, t);
}

// One group: quads taken from LDS where q < NLDS, else from the SGPR record
//   SGPR record: cx0 cx1 cy0 cy1 cz0 cz1 k0 k1 | cx2 cx3 cy2 cy3 cz2 cz3 k2 k3
//   LDS record (NLDS quads of 16 B): -k[4], cx[4], cy[4], cz[4]
static __device__ __forceinline__ void push(uint32_t* slot, uint32_t& n, bool pass, int j) {
  if (pass) slot[(n & 7u) * 64u] = (uint32_t)j;
  n += pass ? 1u : 0u;
}

template <int NLDS>
static __device__ __forceinline__ void group_test(const f16v g, const f4v* q, const Ray& r, uint32_t* slot, uint32_t& n, int j) {
  const f2v cx0 = NLDS > 1 ? q[1].xy : g.s01, cx1 = NLDS > 1 ? q[1].zw : g.s89;
  const f2v cy0 = NLDS > 2 ? q[2].xy : g.s23, cy1 = NLDS > 2 ? q[2].zw : g.sab;
  const f2v cz0 = NLDS > 3 ? q[3].xy : g.s45, cz1 = NLDS > 3 ? q[3].zw : g.scd;
  const f2v f0 = pair_filter(cx0, cy0, cz0, q[0].xy, r);
  const f2v f1 = pair_filter(cx1, cy1, cz1, q[0].zw, r);
  const float mx = fmaxf(fmaxf(f0.x, f0.y), fmaxf(f1.x, f1.y));
  if (mx >= r.threshold) {   // rare: survivors go onto a per-lane stack in LDS, as in the renderer
    push(slot, n, f0.x >= r.threshold, j);
    push(slot, n, f0.y >= r.threshold, j + 1);
    push(slot, n, f1.x >= r.threshold, j + 2);
    push(slot, n, f1.y >= r.threshold, j + 3);
  }
}

template <int NLDS>
__global__ void __launch_bounds__(kThreads, 5) filter_kernel(const float* sgpr_table, const float* lds_table,
                                                              uint32_t* out, int rays) {
  extern __shared__ f4v lds[];
  for (int i = threadIdx.x; i < (kGroups + 2) * NLDS; i += kThreads) {
    const int g = i / NLDS, qd = i % NLDS;
    lds[i] = ((const f4v*)lds_table)[g * 4 + qd];
  }
  __syncthreads();
  const geom_ptr geom = (geom_ptr)(uintptr_t)sgpr_table;
  const uint32_t tid = blockIdx.x * kThreads + threadIdx.x;
  uint32_t* const slot = (uint32_t*)(lds + (kGroups + 2) * 4) + (threadIdx.x >> 6) * 512 + (threadIdx.x & 63);
  uint32_t hits = 0;
  float ph = (float)(tid & 1023u) * 0.006f;
  for (int it = 0; it < rays; ++it) {
    // a ray from around the camera position towards the scene; differs per lane and per trip
    ph += 0.37f;
    const float ox = 13.0f + 0.01f * __sinf(ph), oy = 2.0f + 0.01f * __sinf(0.3f * ph), oz = 3.0f + 0.01f * __cosf(ph);
    float dx = -13.0f + 4.0f * __sinf(1.7f * ph), dy = -2.0f + 1.5f * __cosf(2.3f * ph), dz = -3.0f + 4.0f * __cosf(1.1f * ph);
    const float il = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= il; dy *= il; dz *= il;
    Ray r;
    r.ux = f2v{dx, dx}; r.uy = f2v{dy, dy}; r.uz = f2v{dz, dz};
    const float nou = -(ox * dx + oy * dy + oz * dz);
    r.nou = f2v{nou, nou};
    r.ox2 = f2v{2 * ox, 2 * ox}; r.oy2 = f2v{2 * oy, 2 * oy}; r.oz2 = f2v{2 * oz, 2 * oz};
    const float oo = ox * ox + oy * oy + oz * oz;
    r.threshold = oo - 0x1p-18f * (oo + 16.0f * 16.0f);

    uint32_t addr = (uint32_t)(uintptr_t)(lds_ptr)lds;
    asm volatile("" : "+v"(addr));
    lds_ptr q = (lds_ptr)(uintptr_t)addr;
    f16v ga = geom[0];
    f4v qa[4], qb[4];
#pragma unroll
    for (int k = 0; k < NLDS; ++k) qa[k] = q[k];
    uint32_t n = 0;
    for (int g = 0; g < kGroups; g += 2) {
      const f16v gb = geom[g + 1];
#pragma unroll
      for (int k = 0; k < NLDS; ++k) qb[k] = q[NLDS + k];
      group_test<NLDS>(ga, qa, r, slot, n, 4 * g);
      ga = geom[g + 2];
#pragma unroll
      for (int k = 0; k < NLDS; ++k) qa[k] = q[2 * NLDS + k];
      q += 2 * NLDS;
      group_test<NLDS>(gb, qb, r, slot, n, 4 * g + 4);
    }
    hits += n + slot[0];
  }
  out[tid] = hits;
}

template <int NLDS>
static void run(const float* d_sgpr, const float* d_lds, uint32_t* d_out, int blocks, int rays, double clock_hz,
                std::vector<uint32_t>& host_out) {
  // 30 KB of LDS per block: exactly five blocks fit a CU, as in the renderer (whose VGPR count sets
  // that limit); with less the dispatcher packs up to eight blocks onto some CUs and the slowest CU
  // decides the time
  const size_t lds_bytes = 30 * 1024;
  static_assert(sizeof(f4v) * (kGroups + 2) * 4 + 4 * 512 * 4 <= 30 * 1024, "table must fit the LDS allocation");
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(filter_kernel<NLDS>, dim3(blocks), dim3(kThreads), lds_bytes, 0, d_sgpr, d_lds, d_out, rays);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  CHECK(hipMemcpy(host_out.data(), d_out, host_out.size() * 4, hipMemcpyDeviceToHost));
  uint64_t total = 0;
  for (uint32_t v : host_out) total += v;
  const double waves = (double)blocks * kThreads / 64.0;
  const double simd_cycles = best * 1e-3 * clock_hz * 1024.0;   // 256 CUs x 4 SIMDs
  printf("quads from LDS %d (SGPR %d): %8.3f ms   %6.2f SIMD cycles per group and wave   (%6.2f per sphere)   checksum %llu\n",
         NLDS, 4 - NLDS, best, simd_cycles / (waves * rays * kGroups), simd_cycles / (waves * rays * kGroups * 4.0),
         (unsigned long long)total);
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const int rays = argc > 1 ? atoi(argv[1]) : 400;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 5;   // 5 blocks of 4 waves per CU = 5 waves per SIMD
  const double clock_hz = argc > 2 ? atof(argv[2]) * 1e6 : 2.36e9;
  printf("device %s  CUs %d  blocks %d x %d threads  rays/lane %d  clock assumed %.0f MHz\n", prop.name, cus, blocks, kThreads,
         rays, clock_hz * 1e-6);
  if (cus * 4 != 1024) printf("note: cycles per group assume 1024 SIMDs\n");

  // the book scene's lattice of small spheres plus padding (k = +inf never passes)
  std::vector<float> sg((kGroups + 2) * 16), ld((kGroups + 2) * 16);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
  for (int g = 0; g < kGroups + 2; ++g) {
    float cx[4], cy[4], cz[4], k[4];
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * g + i;
      if (g < kGroups - 1) {
        cx[i] = (float)(j / 22 - 11) + 0.9f * rnd(); cy[i] = 0.2f; cz[i] = (float)(j % 22 - 11) + 0.9f * rnd();
        k[i] = cx[i] * cx[i] + cy[i] * cy[i] + cz[i] * cz[i] - 0.04f;
      } else { cx[i] = cy[i] = cz[i] = 0.0f; k[i] = INFINITY; }
    }
    float* a = &sg[g * 16];
    a[0] = cx[0]; a[1] = cx[1]; a[2] = cy[0]; a[3] = cy[1]; a[4] = cz[0]; a[5] = cz[1]; a[6] = k[0]; a[7] = k[1];
    a[8] = cx[2]; a[9] = cx[3]; a[10] = cy[2]; a[11] = cy[3]; a[12] = cz[2]; a[13] = cz[3]; a[14] = k[2]; a[15] = k[3];
    float* b = &ld[g * 16];
    for (int i = 0; i < 4; ++i) { b[i] = -k[i]; b[4 + i] = cx[i]; b[8 + i] = cy[i]; b[12 + i] = cz[i]; }
  }
  float *d_sgpr, *d_lds; uint32_t* d_out;
  CHECK(hipMalloc(&d_sgpr, sg.size() * 4)); CHECK(hipMalloc(&d_lds, ld.size() * 4));
  CHECK(hipMalloc(&d_out, (size_t)blocks * kThreads * 4));
  CHECK(hipMemcpy(d_sgpr, sg.data(), sg.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_lds, ld.data(), ld.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> host_out((size_t)blocks * kThreads);
  run<1>(d_sgpr, d_lds, d_out, blocks, rays, clock_hz, host_out);
  run<2>(d_sgpr, d_lds, d_out, blocks, rays, clock_hz, host_out);
  run<3>(d_sgpr, d_lds, d_out, blocks, rays, clock_hz, host_out);
  run<4>(d_sgpr, d_lds, d_out, blocks, rays, clock_hz, host_out);
  CHECK(hipFree(d_sgpr)); CHECK(hipFree(d_lds)); CHECK(hipFree(d_out));
  return 0;
}
