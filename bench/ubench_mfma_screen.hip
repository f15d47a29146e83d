// Feasibility probe: the sphere screen (discriminant >= -margin) as ONE bf16 dot product per
// (ray, sphere) on the matrix pipe of gfx950.
//   disc = sum_{a<=b} Q_ab(c) U_ab(u) + c.w - k - |o_perp|^2,   Q = (cx^2, cy^2, cz^2, 2cxcy, 2cxcz, 2cycz),
//   U = (ux^2, uy^2, uz^2, uxuy, uxuz, uyuz),  w = 2 (o - (o.u) u),  k = |c|^2 - r^2
// Ten binary32 terms; every operand is split in three bf16 pieces and the six largest piece products
// of every term are kept: 60 products = two v_mfma_f32_16x16x32_bf16 per tile of 16 spheres x 16 rays.
// Reports (1) the error of the matrix-pipe value against binary64, in units of 2^-18 (|o|^2 + cmax^2)
// (the margin of the VALU screen), (2) matrix-pipe cycles per sphere and wave with the compare.
// Build: hipcc -O3 --offload-arch=gfx950 bench/ubench_mfma_screen.hip -o bench/ubench_mfma_screen
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline uint16_t bf16_rn_bits(float x) {
  uint32_t u; memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_float(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(double x, uint16_t p[3]) {
  double rest = x;
  for (int i = 0; i < 3; ++i) { p[i] = bf16_rn_bits((float)rest); rest -= (double)bf16_to_float(p[i]); }
}
__device__ __forceinline__ uint32_t d_bf16_rn_bits(float x) {
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ void d_split3(float x, uint32_t p[3]) {
  float rest = x;
#pragma unroll
  for (int i = 0; i < 3; ++i) { p[i] = d_bf16_rn_bits(rest); rest -= __uint_as_float(p[i] << 16); }
}

// slot 6 t + q of term t: sphere piece {1,1,2,1,2,3}[q] times ray piece {1,2,1,3,2,1}[q]
__constant__ int kRayPiece[6] = {0, 1, 0, 2, 1, 0};
static const int kSpherePiece[6] = {0, 0, 1, 0, 1, 2};

struct Ray { float ox, oy, oz, ux, uy, uz; };

// One wave = 64 rays.  table: per batch of 16 spheres, per MFMA m (0, 1), per lane: 8 bf16 (uint4).
// out (optional): value per (ray, sphere) for the first `n_out_rays` rays.  survivors: total count.
template <bool WRITE_OUT>
__global__ void __launch_bounds__(256) screen_kernel(const Ray* rays, const uint4* table, int n_batches, float cmax2,
                                                     float margin_scale, float* out, int n_out_rays, int n_spheres,
                                                     unsigned long long* survivors, int iters, unsigned long long* clk) {
  __shared__ uint16_t stage[4][64][64];   // per wave: 64 rays x 64 slots
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_id = (blockIdx.x * 4 + wave) * 64 + lane;
  const Ray r = rays[ray_id];
  // ray terms
  const float ou = r.ox * r.ux + r.oy * r.uy + r.oz * r.uz;
  const float wx = 2.0f * (r.ox - ou * r.ux), wy = 2.0f * (r.oy - ou * r.uy), wz = 2.0f * (r.oz - ou * r.uz);
  const float term[10] = {r.ux * r.ux, r.uy * r.uy, r.uz * r.uz, r.ux * r.uy, r.ux * r.uz, r.uy * r.uz, wx, wy, wz, 1.0f};
  const float oo = r.ox * r.ox + r.oy * r.oy + r.oz * r.oz;
  const float operp2 = oo - ou * ou;
  const float threshold = operp2 - margin_scale * (1.0f / 262144.0f) * (oo + cmax2);
#pragma unroll
  for (int t = 0; t < 10; ++t) {
    uint32_t p[3];
    d_split3(term[t], p);
#pragma unroll
    for (int q = 0; q < 6; ++q) stage[wave][lane][6 * t + q] = (uint16_t)p[kRayPiece[q]];
  }
#pragma unroll
  for (int s = 60; s < 64; ++s) stage[wave][lane][s] = 0;
  __syncthreads();
  // B fragments: group g (rays 16 g .. 16 g + 15), MFMA m: slots 32 m + 8 (lane >> 4) .. + 7 of ray 16 g + (lane & 15)
  bf16x8 B[4][2];
  float thr[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int src = 16 * g + (lane & 15);
#pragma unroll
    for (int m = 0; m < 2; ++m) B[g][m] = *(const bf16x8*)&stage[wave][src][32 * m + 8 * (lane >> 4)];
    thr[g] = __shfl(threshold, src, 64);
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned int count = 0;
  for (int it = 0; it < iters; ++it) {
    uint4 n0 = table[lane], n1 = table[64 + lane];   // operands of batch 0; the loop fetches one batch ahead
    for (int b = 0; b < n_batches; ++b) {
      const uint4 a0 = n0, a1 = n1;
      const int nb = b + 1 < n_batches ? b + 1 : 0;
      n0 = table[(size_t)(2 * nb) * 64 + lane]; n1 = table[(size_t)(2 * nb + 1) * 64 + lane];
      bf16x8 A0, A1;
      memcpy(&A0, &a0, 16); memcpy(&A1, &a1, 16);
      f32x4 accs[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) { const f32x4 z = {0, 0, 0, 0}; accs[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B[g][0], z, 0, 0, 0); }
#pragma unroll
      for (int g = 0; g < 4; ++g) accs[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B[g][1], accs[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 acc = accs[g];
        // four compares and scalar ORs: fmaxf on matrix results would first canonicalise every operand
        if ((acc[0] >= thr[g]) | (acc[1] >= thr[g]) | (acc[2] >= thr[g]) | (acc[3] >= thr[g])) {
#pragma unroll
          for (int q = 0; q < 4; ++q) count += acc[q] >= thr[g] ? 1u : 0u;
        }
        if (WRITE_OUT && it == 0) {
          const int ray = (blockIdx.x * 4 + wave) * 64 + 16 * g + (lane & 15);
          if (ray < n_out_rays)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int sph = 16 * b + 4 * (lane >> 4) + q;
              if (sph < n_spheres) out[(size_t)ray * n_spheres + sph] = acc[q];
            }
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 3 && threadIdx.x == 0) clk[0] = t1 - t0;
  for (int off = 32; off > 0; off >>= 1) count += __shfl_xor(count, off, 64);
  if (lane == 0) atomicAdd(survivors, (unsigned long long)count);
}

// Second probe: the screen as the renderer would run it.  One 1024-thread workgroup per CU (four
// waves per SIMD); the operand table sits in LDS; every "segment" rebuilds the ray fragments (staged
// 16 rays at a time through 2 KB of LDS per wave) and hands every survivor from the lane that holds
// the result to the stack of the lane that owns the ray (LDS atomic on the owner's counter).
constexpr int kScanWaves = 16, kDepth = 12;
__global__ void __launch_bounds__(1024) scan_kernel(const Ray* rays, const uint4* table, int n_batches, float cmax2,
                                                    unsigned long long* survivors, int iters, unsigned long long* clk) {
  extern __shared__ uint4 lds[];
  uint4* const tab = lds;                                             // n_batches * 2 * 64
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* const stage = tab + (size_t)n_batches * 128 + wave * 128;    // 16 rays x 8 uint4 (2 KB)
  uint32_t* const wave_words = (uint32_t*)(tab + (size_t)n_batches * 128 + kScanWaves * 128) + wave * (kDepth + 1) * 64;
  uint32_t* const cnt = wave_words;                                   // 64 counters
  uint32_t* const stack = wave_words + 64;                            // [kDepth][64]
  for (int i = threadIdx.x; i < n_batches * 128; i += 1024) tab[i] = table[i];
  __syncthreads();
  const Ray r = rays[(blockIdx.x * kScanWaves + wave) * 64 + lane];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned int total = 0;
  for (int it = 0; it < iters; ++it) {
    // ---- per segment: ray terms, fragments, thresholds
    const float ox = r.ox + 1e-6f * (float)it;   // keeps the compiler from hoisting the segment set-up
    const float ou = ox * r.ux + r.oy * r.uy + r.oz * r.uz;
    const float wx = 2.0f * (ox - ou * r.ux), wy = 2.0f * (r.oy - ou * r.uy), wz = 2.0f * (r.oz - ou * r.uz);
    const float term[10] = {r.ux * r.ux, r.uy * r.uy, r.uz * r.uz, r.ux * r.uy, r.ux * r.uz, r.uy * r.uz, wx, wy, wz, 1.0f};
    const float oo = ox * ox + r.oy * r.oy + r.oz * r.oz;
    const float threshold = (oo - ou * ou) - (1.0f / 262144.0f) * (oo + cmax2);
    uint32_t slot_words[32];   // 64 bf16 slots, two to a word
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      uint32_t p[3];
      d_split3(term[t], p);
      // slots 6t .. 6t+5 hold ray pieces {1,2,1,3,2,1}: words 3t .. 3t+2
      slot_words[3 * t] = p[0] | (p[1] << 16);
      slot_words[3 * t + 1] = p[0] | (p[2] << 16);
      slot_words[3 * t + 2] = p[1] | (p[0] << 16);
    }
    slot_words[30] = 0; slot_words[31] = 0;
    bf16x8 B[4][2];
    float thr[4];
    cnt[lane] = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if ((lane >> 4) == g) {
#pragma unroll
        for (int w = 0; w < 8; ++w)
          stage[(lane & 15) * 8 + w] = make_uint4(slot_words[4 * w], slot_words[4 * w + 1], slot_words[4 * w + 2], slot_words[4 * w + 3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const uint4 v = stage[(lane & 15) * 8 + 4 * m + (lane >> 4)];   // slots 32 m + 8 (lane >> 4) .. + 7
        memcpy(&B[g][m], &v, 16);
      }
      thr[g] = __shfl(threshold, 16 * g + (lane & 15), 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    // ---- scan
    uint4 n0 = tab[lane], n1 = tab[64 + lane];
    for (int b = 0; b < n_batches; ++b) {
      const uint4 a0 = n0, a1 = n1;
      const int nb = b + 1 < n_batches ? b + 1 : 0;
      n0 = tab[(size_t)(2 * nb) * 64 + lane]; n1 = tab[(size_t)(2 * nb + 1) * 64 + lane];
      bf16x8 A0, A1;
      memcpy(&A0, &a0, 16); memcpy(&A1, &a1, 16);
      f32x4 accs[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) { const f32x4 z = {0, 0, 0, 0}; accs[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B[g][0], z, 0, 0, 0); }
#pragma unroll
      for (int g = 0; g < 4; ++g) accs[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B[g][1], accs[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (accs[g][q] >= thr[g]) {   // rare per (g, q): the owner's stack takes the list index
            const uint32_t owner = 16u * g + (lane & 15u);
            const uint32_t pos = atomicAdd(&cnt[owner], 1u);
            if (pos < (uint32_t)kDepth) stack[pos * 64u + owner] = 16u * b + 4u * (lane >> 4) + q;
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t mine = cnt[lane];
    total += mine;
    if (mine) total += stack[lane] & 0u;   // the stack is read by its owner
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 3 && threadIdx.x == 0) clk[0] = t1 - t0;
  for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off, 64);
  if (lane == 0) atomicAdd(survivors, (unsigned long long)total);
}

int main() {
  // a scene shaped like the reference's: 22 x 22 grid of r = 0.2 spheres plus three of r = 1
  std::vector<double> cx, cy, cz, rr;
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) { cx.push_back(a + 0.9 * rnd()); cy.push_back(0.2); cz.push_back(b + 0.9 * rnd()); rr.push_back(0.2); }
  const double big[3][3] = {{0, 1, 0}, {-4, 1, 0}, {4, 1, 0}};
  for (auto& p : big) { cx.push_back(p[0]); cy.push_back(p[1]); cz.push_back(p[2]); rr.push_back(1.0); }
  const int n = (int)cx.size(), n_batches = (n + 15) / 16;
  double cmax = 0;
  for (int j = 0; j < n; ++j) cmax = fmax(cmax, sqrt(cx[j] * cx[j] + cy[j] * cy[j] + cz[j] * cz[j]) + rr[j]);
  // operand table
  std::vector<uint16_t> slots((size_t)n_batches * 16 * 64, 0);
  for (int j = 0; j < n_batches * 16; ++j) {
    double t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, -1e30};   // padding: k = +huge, never passes
    if (j < n) {
      const double x = cx[j], y = cy[j], z = cz[j];
      const double q[10] = {x * x, y * y, z * z, 2 * x * y, 2 * x * z, 2 * y * z, x, y, z, -(x * x + y * y + z * z - rr[j] * rr[j])};
      memcpy(t, q, sizeof(q));
    }
    for (int k = 0; k < 10; ++k) {
      uint16_t p[3];
      split3(t[k], p);
      for (int q = 0; q < 6; ++q) slots[(size_t)j * 64 + 6 * k + q] = p[kSpherePiece[q]];
    }
  }
  std::vector<uint16_t> table((size_t)n_batches * 2 * 64 * 8);
  for (int b = 0; b < n_batches; ++b)
    for (int m = 0; m < 2; ++m)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 8; ++e)
          table[(((size_t)(2 * b + m)) * 64 + l) * 8 + e] = slots[(size_t)(16 * b + (l & 15)) * 64 + 32 * m + 8 * (l >> 4) + e];
  // rays: a third from the book's camera, the rest leaving random sphere surfaces
  const int n_blocks = 256 * 4, n_rays = n_blocks * 256;   // the timing grids below use at most n_blocks workgroups
  std::vector<Ray> rays(n_rays);
  for (int i = 0; i < n_rays; ++i) {
    double o[3], d[3];
    if (i % 3 == 0) { o[0] = 13 + 0.05 * (rnd() - 0.5); o[1] = 2 + 0.05 * (rnd() - 0.5); o[2] = 3; d[0] = -13 + 8 * (rnd() - 0.5); d[1] = -2 + 5 * (rnd() - 0.5); d[2] = -3 + 8 * (rnd() - 0.5); }
    else {
      const int j = (int)(rnd() * n) % n;
      double v[3]; double l;
      do { for (double& c : v) c = 2 * rnd() - 1; l = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]; } while (l >= 1 || l < 1e-4);
      l = sqrt(l);
      o[0] = cx[j] + rr[j] * v[0] / l; o[1] = cy[j] + rr[j] * v[1] / l; o[2] = cz[j] + rr[j] * v[2] / l;
      do { for (double& c : d) c = 2 * rnd() - 1; l = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]; } while (l >= 1 || l < 1e-4);
    }
    const double l = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    rays[i] = {(float)o[0], (float)o[1], (float)o[2], (float)(d[0] / l), (float)(d[1] / l), (float)(d[2] / l)};
    // renormalise in binary32 like the kernel's unit directions
    Ray& r = rays[i];
    const float fl = sqrtf(r.ux * r.ux + r.uy * r.uy + r.uz * r.uz);
    r.ux /= fl; r.uy /= fl; r.uz /= fl;
  }
  Ray* d_rays; uint4* d_table; float* d_out; unsigned long long *d_surv, *d_clk;
  const int n_out = 4096;
  CK(hipMalloc(&d_rays, sizeof(Ray) * n_rays)); CK(hipMalloc(&d_table, table.size() * 2));
  CK(hipMalloc(&d_out, sizeof(float) * (size_t)n_out * n)); CK(hipMalloc(&d_surv, 8)); CK(hipMalloc(&d_clk, 8));
  CK(hipMemcpy(d_rays, rays.data(), sizeof(Ray) * n_rays, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_table, table.data(), table.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemset(d_surv, 0, 8));
  const float cmax2 = (float)(cmax * cmax);
  screen_kernel<true><<<n_blocks, 256>>>(d_rays, d_table, n_batches, cmax2, 1.0f, d_out, n_out, n, d_surv, 1, d_clk);
  CK(hipDeviceSynchronize());
  std::vector<float> out((size_t)n_out * n);
  CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long surv = 0;
  CK(hipMemcpy(&surv, d_surv, 8, hipMemcpyDeviceToHost));
  // error against binary64, in margins
  double worst = 0, worst_signed_low = 0;
  unsigned long long true_hits = 0, dropped = 0, passed = 0;
  for (int i = 0; i < n_out; ++i) {
    const Ray& r = rays[i];
    const double oo = (double)r.ox * r.ox + (double)r.oy * r.oy + (double)r.oz * r.oz;
    const double ou = (double)r.ox * r.ux + (double)r.oy * r.uy + (double)r.oz * r.uz;
    const double margin = (oo + cmax * cmax) / 262144.0;
    for (int j = 0; j < n; ++j) {
      const double ocx = r.ox - cx[j], ocy = r.oy - cy[j], ocz = r.oz - cz[j];
      const double b = ocx * r.ux + ocy * r.uy + ocz * r.uz;
      const double disc = b * b - (ocx * ocx + ocy * ocy + ocz * ocz - rr[j] * rr[j]);
      const double got = (double)out[(size_t)i * n + j] - (oo - ou * ou);   // the matrix value minus |o_perp|^2 = disc
      const double err = (got - disc) / margin;
      if (fabs(err) > worst) worst = fabs(err);
      if (err < worst_signed_low) worst_signed_low = err;
      if (disc >= 0) { ++true_hits; if (got < -margin) ++dropped; }
      if (got >= -margin) ++passed;
    }
  }
  printf("spheres %d (%d batches of 16), rays checked %d: |matrix value - binary64 disc| <= %.4f margins (most negative %.4f);\n"
         "  with margin x1: %llu of %llu true hits dropped, %llu pairs pass (%.3f per ray); kernel-wide survivors per ray %.3f\n",
         n, n_batches, n_out, worst, worst_signed_low, dropped, true_hits, passed, (double)passed / n_out, (double)surv / n_rays);
  // timing
  for (int blocks_per_cu : {2, 3, 4}) {
    const int grid = 256 * blocks_per_cu, iters = 200;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    screen_kernel<false><<<grid, 256>>>(d_rays, d_table, n_batches, cmax2, 1.0f, nullptr, 0, n, d_surv, 20, d_clk);
    CK(hipEventRecord(e0));
    screen_kernel<false><<<grid, 256>>>(d_rays, d_table, n_batches, cmax2, 1.0f, nullptr, 0, n, d_surv, iters, d_clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk; CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    // per SIMD: blocks_per_cu waves, each iters scans of n_batches*16 spheres
    const double simd_cycles_per_sphere_wave = (double)clk / ((double)iters * n_batches * 16) / blocks_per_cu;
    printf("waves/SIMD %d: %.3f ms, one wave %.1f cycles per sphere (wall), %.2f SIMD cycles per sphere and wave "
           "(VALU screen today: 16.4)\n", blocks_per_cu, ms, (double)clk / ((double)iters * n_batches * 16), simd_cycles_per_sphere_wave);
  }
  {
    const size_t lds_bytes = (size_t)n_batches * 128 * 16 + kScanWaves * 128 * 16 + (size_t)kScanWaves * (kDepth + 1) * 64 * 4;
    CK(hipFuncSetAttribute((const void*)scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    const int grid = 256, iters = 50;   // one workgroup of 16 waves per CU; rays 0 .. 256*1024-1
    CK(hipMemset(d_surv, 0, 8));
    scan_kernel<<<grid, 1024, lds_bytes>>>(d_rays, d_table, n_batches, cmax2, d_surv, 1, d_clk);
    CK(hipDeviceSynchronize());
    unsigned long long s1 = 0; CK(hipMemcpy(&s1, d_surv, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    scan_kernel<<<grid, 1024, lds_bytes>>>(d_rays, d_table, n_batches, cmax2, d_surv, iters, d_clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk; CK(hipMemcpy(&clk, d_clk, 8, hipMemcpyDeviceToHost));
    printf("renderer-shaped scan (LDS operands %zu KB of LDS per CU, fragments rebuilt per segment, survivors handed to their owners):\n"
           "  %.3f ms, survivors per ray %.3f (first probe: %.3f), one wave %.1f cycles per sphere (wall), %.2f SIMD cycles per sphere "
           "and wave at 4 waves/SIMD (VALU screen today: 16.4)\n", lds_bytes / 1024, ms, (double)s1 / (grid * 1024.0), (double)surv / n_rays,
           (double)clk / ((double)iters * n_batches * 16), (double)clk / ((double)iters * n_batches * 16) / 4.0);
  }
  return 0;
}
