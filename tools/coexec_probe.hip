// Probe (gfx950): do v_mfma_f32_16x16x4_f32 and plain VALU instructions overlap on one SIMD?
//   A. two waves per SIMD, one issuing only MFMAs, the other only v_fma_f32: wall time vs each alone
//   B. one wave per SIMD issuing 1 MFMA + k independent v_fma_f32 per iteration, k = 0..12
// Diagnostic tool, not part of libampconv.so.  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

#define VFMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(x))

// role 0: MFMA wave, role 1: VALU wave.  mask: bit 0 = waves 0..3 run, bit 1 = waves 4..7 run
__global__ __launch_bounds__(512) void two_waves(float *sink, unsigned long long *cyc, int iters, float x, int mask,
                                                 int role_lo, int role_hi) {
  const int wave = threadIdx.x >> 6;
  const bool hi = wave >= 4;
  if (!((mask >> (hi ? 1 : 0)) & 1)) return;
  const int role = hi ? role_hi : role_lo;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float a = x + threadIdx.x, b = x - threadIdx.x, m = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  const unsigned long long t0 = now();
  if (role == 0) {
    for (int i = 0; i < iters; ++i) {
      c0 = MF16(a, b, c0); c1 = MF16(a, b, c1); c2 = MF16(a, b, c2); c3 = MF16(a, b, c3);
    }
  } else {
    for (int i = 0; i < iters; ++i) {      // 32 independent-ish VALU instructions (8 chains) per iteration
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        VFMA(v0); VFMA(v1); VFMA(v2); VFMA(v3); VFMA(v4); VFMA(v5); VFMA(v6); VFMA(v7);
      }
    }
  }
  const unsigned long long t1 = now();
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

template <int K>
__global__ __launch_bounds__(256) void one_wave(float *sink, unsigned long long *cyc, int iters, float x) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  float a = x + threadIdx.x, b = x - threadIdx.x, m = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
  const unsigned long long t0 = now();
  for (int i = 0; i < iters; ++i) {
    c0 = MF16(a, b, c0);
#pragma unroll
    for (int j = 0; j < K / 4; ++j) { VFMA(v0); VFMA(v1); VFMA(v2); VFMA(v3); }
    c1 = MF16(a, b, c1);
#pragma unroll
    for (int j = 0; j < K / 4; ++j) { VFMA(v0); VFMA(v1); VFMA(v2); VFMA(v3); }
  }
  const unsigned long long t1 = now();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  sink[threadIdx.x] = c0[0] + c1[1] + v0 + v1 + v2 + v3;
}

static float *sink;
static unsigned long long *cyc;

static void run_two(const char *what, int mask, int role_lo, int role_hi) {
  const int iters = 20000;
  hipMemset(cyc, 0, 8 * sizeof(unsigned long long));
  two_waves<<<1, 512>>>(sink, cyc, iters, 1.f, mask, role_lo, role_hi);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(8);
  hipMemcpy(h.data(), cyc, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  printf("%-58s", what);
  for (int w = 0; w < 8; ++w) printf(" %7.1f", (double)h[w] / iters);
  printf("   (s_memtime ticks per iteration = 4 MFMAs / 32 v_fma, waves 0..7)\n");
}

template <int K>
static void run_one() {
  const int iters = 20000;
  one_wave<K><<<1, 256>>>(sink, cyc, iters, 1.f);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("one wave per SIMD, per MFMA + %2d v_fma: %7.2f ticks\n", K, (double)h[0] / iters / 2);
}

int main() {
  hipMalloc(&sink, 4096);
  hipMalloc(&cyc, 64);
  run_two("waves 0..3 MFMA only", 1, 0, 0);
  run_two("waves 0..3 VALU only", 1, 1, 1);
  run_two("waves 0..3 MFMA + waves 4..7 MFMA", 3, 0, 0);
  run_two("waves 0..3 VALU + waves 4..7 VALU", 3, 1, 1);
  run_two("waves 0..3 MFMA + waves 4..7 VALU (same SIMDs)", 3, 0, 1);
  run_one<0>(); run_one<4>(); run_one<8>(); run_one<12>(); run_one<16>(); run_one<24>();
  return 0;
}
