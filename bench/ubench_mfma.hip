// Micro-benchmarks for running the sphere filter on the matrix pipe of gfx950.
//   1. operand / result lane maps of the f32 multi-block MFMAs (K = 1 forms)
//   2. how v_mfma_f32_32x32x1_2b_f32 chains overlap with packed FP32 VALU work, inside one wave
//      and across the waves of a SIMD
// Build: hipcc -O3 --offload-arch=gfx950 bench/ubench_mfma.hip -o bench/ubench_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void map_kernel(float* out) {
  const int l = threadIdx.x;
  f32x32 z32 = {0};
  f32x16 z16 = {0};
  f32x4 z4 = {0};
  // which A element / which B element feeds each result register
  f32x32 a32 = __builtin_amdgcn_mfma_f32_32x32x1f32((float)l, 1.0f, z32, 0, 0, 0);
  f32x32 b32 = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, (float)l, z32, 0, 0, 0);
  f32x16 a16 = __builtin_amdgcn_mfma_f32_16x16x1f32((float)l, 1.0f, z16, 0, 0, 0);
  f32x16 b16 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)l, z16, 0, 0, 0);
  f32x4 a4 = __builtin_amdgcn_mfma_f32_4x4x1f32((float)l, 1.0f, z4, 0, 0, 0);
  f32x4 b4 = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)l, z4, 0, 0, 0);
  for (int r = 0; r < 32; ++r) { out[(0 * 64 + l) * 32 + r] = a32[r]; out[(1 * 64 + l) * 32 + r] = b32[r]; }
  for (int r = 0; r < 16; ++r) { out[(2 * 64 + l) * 32 + r] = a16[r]; out[(3 * 64 + l) * 32 + r] = b16[r]; }
  for (int r = 0; r < 4; ++r) { out[(4 * 64 + l) * 32 + r] = a4[r]; out[(5 * 64 + l) * 32 + r] = b4[r]; }
}

// unrelated VALU work: KIND 0 packed fma, 1 plain v_fma_f32, 2 integer xor-add
template <int KIND, int EXTRA>
static __device__ __forceinline__ void extra_work(f2v (&e)[8], f2v m) {
#pragma unroll
  for (int k = 0; k < EXTRA; ++k) {
    if (KIND == 0) e[k & 7] = __builtin_elementwise_fma(e[k & 7], m, m);
    else if (KIND == 1) e[k & 7].x = fmaf(e[k & 7].x, m.x, m.y);
    else e[k & 7].x = __uint_as_float((__float_as_uint(e[k & 7].x) ^ __float_as_uint(m.x)) + __float_as_uint(m.y));
  }
}

// One "batch": 7 MFMAs (two accumulation chains, 3 + 4 deep) and `post` packed fmas + maxes on the
// results, plus EXTRA independent packed fmas standing for unrelated VALU work.
template <int MFMA, int POST, int EXTRA, int KIND>
__global__ void __launch_bounds__(256) overlap_kernel(const float* in, float* out, int iters) {
  const int l = threadIdx.x & 63;
  float ax = in[l], ay = in[64 + l], az = in[128 + l], ak = in[192 + l];
  float bx = in[256 + l], by = in[320 + l], bz = in[384 + l], one = 1.0f;
  f2v e[8];
  for (int i = 0; i < 8; ++i) e[i] = f2v{in[i + l], in[i + 64 + l]};
  const f2v m = {in[7 + l], in[9 + l]};
  float keep = 0.0f;
  for (int it = 0; it < iters; ++it) {
    f32x32 n = {0}, t = {0};
    if (MFMA) {
      n = __builtin_amdgcn_mfma_f32_32x32x1f32(ax, bx, n, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x1f32(ax, by, t, 0, 0, 0);
      n = __builtin_amdgcn_mfma_f32_32x32x1f32(ay, by, n, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x1f32(ay, bz, t, 0, 0, 0);
      n = __builtin_amdgcn_mfma_f32_32x32x1f32(az, bz, n, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x1f32(az, bx, t, 0, 0, 0);
      t = __builtin_amdgcn_mfma_f32_32x32x1f32(ak, one, t, 0, 0, 0);
    }
    extra_work<KIND, EXTRA>(e, m);
    if (POST) {
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 32; r += 4) {
        f2v n0 = {n[r], n[r + 1]}, t0 = {t[r], t[r + 1]}, n1 = {n[r + 2], n[r + 3]}, t1 = {t[r + 2], t[r + 3]};
        f2v f0 = __builtin_elementwise_fma(n0, n0, t0);
        f2v f1 = __builtin_elementwise_fma(n1, n1, t1);
        mx = fmaxf(mx, fmaxf(fmaxf(f0.x, f0.y), fmaxf(f1.x, f1.y)));
      }
      keep = fmaxf(keep, mx);
    } else if (MFMA) {
      keep += n[0] + t[5];
    }
    ax += 1.0f;  // make the iterations distinct
  }
  float s = keep;
  for (int i = 0; i < 8; ++i) s += e[i].x + e[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// bf16 variant: a batch is 32 spheres x 64 rays through v_mfma_f32_32x32x16_bf16, 4 K-steps (64 slots)
// for each of the two 32-ray halves, then a max tree + compare on the 32 result registers.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
template <int MFMA, int POST, int EXTRA, int KIND>
__global__ void __launch_bounds__(256) overlap_bf16_kernel(const float* in, float* out, int iters) {
  const int l = threadIdx.x & 63;
  bf16x8 a[4], b[8];
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) a[i][k] = (short)(0x3f80 + ((l + i + k) & 31));
  for (int i = 0; i < 8; ++i) for (int k = 0; k < 8; ++k) b[i][k] = (short)(0x3f00 + ((l * 3 + i + k) & 63));
  f2v e[8];
  for (int i = 0; i < 8; ++i) e[i] = f2v{in[i + l], in[i + 64 + l]};
  const f2v m = {in[7 + l], in[9 + l]};
  const float thr = in[300 + l];
  float keep = 0.0f;
  int hits = 0;
  for (int it = 0; it < iters; ++it) {
    f32x16 f0 = {0}, f1 = {0};
    if (MFMA) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], f0, 0, 0, 0);
        f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[4 + k], f1, 0, 0, 0);
      }
    }
    extra_work<KIND, EXTRA>(e, m);
    if (POST) {
      float m0 = f0[0], m1 = f1[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) { m0 = fmaxf(m0, f0[r]); m1 = fmaxf(m1, f1[r]); }
      if (__ballot(m0 >= thr || m1 >= thr) != 0) ++hits;
    } else if (MFMA) {
      keep += f0[0] + f1[5];
    }
    a[0][0] = (short)(a[0][0] ^ (it & 1));  // make the iterations distinct
  }
  float s = keep + (float)hits;
  for (int i = 0; i < 8; ++i) s += e[i].x + e[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// 512-thread blocks put two waves on every SIMD: waves 0-3 only issue MFMAs, waves 4-7 only VALU work.
// MF: 0 none, 1 bf16 32x32x16 (8 per iteration), 2 f32 32x32x1_2b (4 per iteration); same 256 matrix cycles
template <int MF, int KIND, int EXTRA, int PRIO = 0>
__global__ void __launch_bounds__(512) split_kernel(const float* in, float* out, int iters, unsigned long long* clk) {
  const int l = threadIdx.x & 63;
  const bool mfma_role = threadIdx.x < 256;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float s = 0.0f;
  if (mfma_role) {
    if (MF == 1) {
      bf16x8 a[4], b[4];
      for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) { a[i][k] = (short)(0x3f80 + ((l + i + k) & 31)); b[i][k] = (short)(0x3f00 + ((l * 3 + i + k) & 63)); }
      f32x16 f0 = {0}, f1 = {0};
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          f0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[k], f0, 0, 0, 0);
          f1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[3 - k], f1, 0, 0, 0);
        }
      }
      s = f0[0] + f1[3];
    } else if (MF == 2) {
      float ax = in[l], bx = in[64 + l];
      f32x32 n = {0}, t = {0};
      for (int it = 0; it < iters; ++it) {
        n = __builtin_amdgcn_mfma_f32_32x32x1f32(ax, bx, n, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x1f32(bx, ax, t, 0, 0, 0);
        n = __builtin_amdgcn_mfma_f32_32x32x1f32(bx, bx, n, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x1f32(ax, ax, t, 0, 0, 0);
      }
      s = n[0] + t[3];
    }
  } else {
    f2v e[8];
    for (int i = 0; i < 8; ++i) e[i] = f2v{in[i + l], in[i + 64 + l]};
    const f2v m = {in[7 + l], in[9 + l]};
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    for (int it = 0; it < iters; ++it) extra_work<KIND, EXTRA>(e, m);
    for (int i = 0; i < 8; ++i) s += e[i].x + e[i].y;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
  if (blockIdx.x == 7 && threadIdx.x == 256) { clk[2] = t1 - t0; clk[3] = r1 - r0; }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MF, int KIND, int EXTRA, int PRIO = 0>
static void run_split(const char* name, const float* din, float* dout, int iters) {
  unsigned long long* dclk;
  CK(hipMalloc(&dclk, 32));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  iters *= 10;
  split_kernel<MF, KIND, EXTRA, PRIO><<<256, 512>>>(din, dout, iters, dclk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  split_kernel<MF, KIND, EXTRA, PRIO><<<256, 512>>>(din, dout, iters, dclk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[4];
  CK(hipMemcpy(h, dclk, 32, hipMemcpyDeviceToHost));
  const double mhz = (double)h[0] / (double)h[1] * 100.0;
  printf("split %-40s %8.3f ms  clk %5.0f MHz  mfma-role %7.1f  valu-role %7.1f cyc/iteration\n", name, ms, mhz,
         (double)h[0] / iters, (double)h[2] / iters);
  CK(hipFree(dclk));
}

template <int MFMA, int POST, int EXTRA, bool BF16 = false, int KIND = 0>
static void run(const char* name, const float* din, float* dout, int blocks_per_cu, int iters) {
  const int n_cu = 256;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto kern = BF16 ? overlap_bf16_kernel<MFMA, POST, EXTRA, KIND> : overlap_kernel<MFMA, POST, EXTRA, KIND>;
  kern<<<n_cu * blocks_per_cu, 256>>>(din, dout, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<<<n_cu * blocks_per_cu, 256>>>(din, dout, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // cycles per batch per SIMD at 2.4 GHz nominal (each SIMD runs blocks_per_cu waves)
  const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * blocks_per_cu);
  printf("%-34s waves/SIMD %d  %8.3f ms  %7.1f cyc/batch/wave-slot (2.4 GHz)\n", name, blocks_per_cu, ms, cyc);
}

int main(int argc, char** argv) {
  float* dmap;
  CK(hipMalloc(&dmap, 6 * 64 * 32 * sizeof(float)));
  CK(hipMemset(dmap, 0, 6 * 64 * 32 * sizeof(float)));
  map_kernel<<<1, 64>>>(dmap);
  std::vector<float> h(6 * 64 * 32);
  CK(hipMemcpy(h.data(), dmap, h.size() * sizeof(float), hipMemcpyDeviceToHost));
  const char* names[6] = {"32x32x1_2b A-lane", "32x32x1_2b B-lane", "16x16x1_4b A-lane", "16x16x1_4b B-lane",
                          "4x4x1_16b A-lane", "4x4x1_16b B-lane"};
  const int nreg[6] = {32, 32, 16, 16, 4, 4};
  const int lanes[] = {0, 1, 5, 16, 31, 32, 33, 48, 63};
  for (int k = 0; k < 6; ++k) {
    printf("%s (source lane of each result register)\n", names[k]);
    for (int li : lanes) {
      printf("  lane %2d:", li);
      for (int r = 0; r < nreg[k]; ++r) printf(" %2d", (int)h[(k * 64 + li) * 32 + r]);
      printf("\n");
    }
  }

  float *din, *dout;
  std::vector<float> hin(1024);
  for (int i = 0; i < 1024; ++i) hin[i] = 0.25f + 0.001f * (float)(i % 97);
  CK(hipMalloc(&din, 1024 * sizeof(float)));
  CK(hipMemcpy(din, hin.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
  CK(hipMalloc(&dout, 256 * 8 * 256 * sizeof(float)));
  const int iters = 20000;
  run_split<1, 0, 0>("8 bf16 mfma | idle", din, dout, iters);
  run_split<0, 0, 64>("idle | 64 pk_fma", din, dout, iters);
  run_split<1, 0, 64>("8 bf16 mfma | 64 pk_fma", din, dout, iters);
  run_split<0, 1, 100>("idle | 100 v_fma", din, dout, iters);
  run_split<1, 1, 100>("8 bf16 mfma | 100 v_fma", din, dout, iters);
  run_split<0, 2, 50>("idle | 50 int xor+add", din, dout, iters);
  run_split<1, 2, 50>("8 bf16 mfma | 50 int xor+add", din, dout, iters);
  run_split<1, 0, 64, 1>("8 bf16 mfma | 64 pk_fma prio3", din, dout, iters);
  run_split<1, 1, 100, 1>("8 bf16 mfma | 100 v_fma prio3", din, dout, iters);
  run_split<2, 1, 100, 1>("4 f32 mfma | 100 v_fma prio3", din, dout, iters);
  run_split<2, 0, 0>("4 f32 mfma | idle", din, dout, iters);
  run_split<2, 0, 64>("4 f32 mfma | 64 pk_fma", din, dout, iters);
  run_split<2, 1, 100>("4 f32 mfma | 100 v_fma", din, dout, iters);
  run_split<2, 2, 50>("4 f32 mfma | 50 int xor+add", din, dout, iters);
  return 0;
}
