// Probe (gfx950): do v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 and plain VALU instructions overlap on one SIMD,
// (A) from two different waves, (B) inside one wave?  Companion of coexec_probe.hip (which answers "no" for the fp32 MFMA).
// Diagnostic tool, not part of libampconv.so.  hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
#define MFF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define MF4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define VFMA(r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(m), "v"(x))
#define VCVT(r) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(m))
#define VAND(r) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(m))
#define VEXP(r) asm volatile("v_exp_f32 %0, %0" : "+v"(r))

// role 0: 16x16x32 MFMA wave (4 independent accumulators), 1: VALU v_fma, 2: 32x32x16 MFMA wave (2 accumulators),
// 3: v_cvt_pk_bf16_f32, 4: v_and_b32, 5: v_exp_f32, 6: dependent chain of 16x16x32 on ONE accumulator
__global__ __launch_bounds__(512) void two_waves(float *sink, unsigned long long *cyc, int iters, float x, int mask,
                                                 int role_lo, int role_hi) {
  const int wave = threadIdx.x >> 6;
  const bool hi = wave >= 4;
  if (!((mask >> (hi ? 1 : 0)) & 1)) return;
  const int role = __builtin_amdgcn_readfirstlane(hi ? role_hi : role_lo);
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {}, d1 = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(x + threadIdx.x + i); b[i] = (__bf16)(x - i); }
  float m = 1.0001f;
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
  // MFMA roles run 8 x the iterations so that both waves of a SIMD stay busy for a comparable time
  const int n = role == 0 || role == 2 || role == 6 || role >= 7 ? 8 * iters : iters;
  const float fa = x + threadIdx.x, fb = x - threadIdx.x;
  const unsigned long long t0 = now();
#define VALU_LOOP(OP)                                                                            \
  for (int i = 0; i < n; ++i) {                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) { OP(v0); OP(v1); OP(v2); OP(v3); OP(v4); OP(v5); OP(v6); OP(v7); } \
  }
  if (role == 0) {
    for (int i = 0; i < n; ++i) { c0 = MF16(a, b, c0); c1 = MF16(a, b, c1); c2 = MF16(a, b, c2); c3 = MF16(a, b, c3); }
  } else if (role == 6) {
    for (int i = 0; i < n; ++i) { c0 = MF16(a, b, c0); c0 = MF16(a, b, c0); c0 = MF16(a, b, c0); c0 = MF16(a, b, c0); }
  } else if (role == 2) {
    for (int i = 0; i < n; ++i) { d0 = MF32(a, b, d0); d1 = MF32(a, b, d1); }
  } else if (role == 7) {      // fp32 16x16x4, four accumulators
    for (int i = 0; i < n; ++i) { c0 = MFF(fa, fb, c0); c1 = MFF(fa, fb, c1); c2 = MFF(fa, fb, c2); c3 = MFF(fa, fb, c3); }
  } else if (role == 8) {      // fp32 16x16x4, one dependent chain
    for (int i = 0; i < n; ++i) { c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0); }
  } else if (role == 9) {      // fp32 16x16x4 chain with a 4x4x1 on another accumulator in between (the edge kernels' S phase)
    for (int i = 0; i < n; ++i) {
      c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1);
      c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1);
    }
  } else if (role == 11) {     // the same 4 + 4, grouped: four 16x16x4 then four 4x4x1
    for (int i = 0; i < n; ++i) {
      c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0); c0 = MFF(fa, fb, c0);
      c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1);
    }
  } else if (role == 12) {     // 4x4x1 only, one chain (4 per iteration)
    for (int i = 0; i < n; ++i) { c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1); c1 = MF4(fa, fb, c1); }
  } else if (role == 13) {     // 4x4x1 only, two chains alternating
    for (int i = 0; i < n; ++i) { c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2); c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2); }
  } else if (role == 14) {     // 16x16x4 with TWO 4x4x1 behind each (4 + 8)
    for (int i = 0; i < n; ++i) {
      c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2); c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2);
      c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2); c0 = MFF(fa, fb, c0); c1 = MF4(fa, fb, c1); c2 = MF4(fa, fb, c2);
    }
  } else if (role == 10) {     // two fp32 chains alternating
    for (int i = 0; i < n; ++i) { c0 = MFF(fa, fb, c0); c1 = MFF(fa, fb, c1); c0 = MFF(fa, fb, c0); c1 = MFF(fa, fb, c1); }
  } else if (role == 1) {
    VALU_LOOP(VFMA)
  } else if (role == 3) {
    VALU_LOOP(VCVT)
  } else if (role == 4) {
    VALU_LOOP(VAND)
  } else {
    VALU_LOOP(VEXP)
  }
  const unsigned long long t1 = now();
  if ((threadIdx.x & 63) == 0) cyc[wave] = (t1 - t0) / (n / iters);
  sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[5] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

// one wave per SIMD: 1 MFMA (16x16x32 bf16) + K independent v_fma, alternating accumulators
template <int K>
__global__ __launch_bounds__(256) void one_wave(float *sink, unsigned long long *cyc, int iters, float x) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(x + threadIdx.x + i); b[i] = (__bf16)(x - i); }
  float m = 1.0001f;
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
  printf("%-64s", what);
  for (int w = 0; w < 8; ++w) printf(" %7.1f", (double)h[w] / iters);
  printf("\n");
}

template <int K>
static void run_one() {
  const int iters = 20000;
  one_wave<K><<<1, 256>>>(sink, cyc, iters, 1.f);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("one wave per SIMD, per 16x16x32 bf16 MFMA + %2d v_fma: %7.2f ticks\n", K, (double)h[0] / iters / 2);
}

int main() {
  hipMalloc(&sink, 4096);
  hipMalloc(&cyc, 64);
  printf("s_memtime ticks per iteration (4 MFMA16 | 2 MFMA32 | 32 VALU), waves 0..7; waves w and w+4 share SIMD w\n");
  run_two("waves 0..3 MFMA16x16x32 (4 accumulators)", 1, 0, 0);
  run_two("waves 0..3 MFMA16x16x32 (one dependent chain)", 1, 6, 6);
  run_two("waves 0..3 MFMA32x32x16 (2 accumulators)", 1, 2, 2);
  run_two("waves 0..3 fp32 MFMA16x16x4 (4 accumulators)", 1, 7, 7);
  run_two("waves 0..3 fp32 MFMA16x16x4 (one dependent chain)", 1, 8, 8);
  run_two("waves 0..3 fp32 MFMA16x16x4 chain + 4x4x1 between (4 + 4 per iteration)", 1, 9, 9);
  run_two("waves 0..3 fp32 MFMA16x16x4 two chains alternating", 1, 10, 10);
  run_two("waves 0..3 fp32 four 16x16x4 then four 4x4x1 (grouped)", 1, 11, 11);
  run_two("waves 0..3 fp32 4x4x1 one chain (4 per iteration)", 1, 12, 12);
  run_two("waves 0..3 fp32 4x4x1 two chains alternating (4 per iteration)", 1, 13, 13);
  run_two("waves 0..3 fp32 16x16x4 + two 4x4x1 behind each (4 + 8)", 1, 14, 14);
  run_two("fp32 chain + fp32 chain (same SIMDs)", 3, 8, 8);
  run_two("fp32 chain+4x4x1 + same (same SIMDs)", 3, 9, 9);
  run_two("waves 0..3 v_fma", 1, 1, 1);
  run_two("waves 0..3 v_cvt_pk_bf16_f32", 1, 3, 3);
  run_two("waves 0..3 v_and_b32", 1, 4, 4);
  run_two("waves 0..3 v_exp_f32", 1, 5, 5);
  run_two("MFMA16 + MFMA16 (same SIMDs)", 3, 0, 0);
  run_two("v_fma + v_fma (same SIMDs)", 3, 1, 1);
  run_two("v_cvt + v_cvt (same SIMDs)", 3, 3, 3);
  run_two("v_and + v_and (same SIMDs)", 3, 4, 4);
  run_two("v_exp + v_exp (same SIMDs)", 3, 5, 5);
  run_two("MFMA16 + v_fma (same SIMDs)", 3, 0, 1);
  run_two("MFMA16 chain + v_fma (same SIMDs)", 3, 6, 1);
  run_two("MFMA32 + v_fma (same SIMDs)", 3, 2, 1);
  run_two("MFMA16 + v_cvt (same SIMDs)", 3, 0, 3);
  run_two("MFMA16 + v_exp (same SIMDs)", 3, 0, 5);
  // does the fp32-input MFMA of one wave (the edge kernels) co-issue with the bf16 MFMA of another
  // wave (the projections) on the same SIMD, or do the two serialise on the matrix port?
  run_two("fp32 MFMA16x16x4 (lo) + bf16 MFMA16x16x32 (hi) (same SIMDs)", 3, 7, 0);
  run_two("fp32 MFMA16x16x4 (lo) + bf16 MFMA32x32x16 (hi) (same SIMDs)", 3, 7, 2);
  run_two("fp32 16x16x4 chain + 4x4x1 (lo) + bf16 MFMA16x16x32 (hi)", 3, 9, 0);
  run_two("fp32 16x16x4 chain + 4x4x1 (lo) + bf16 MFMA32x32x16 (hi)", 3, 9, 2);
  run_one<0>(); run_one<4>(); run_one<8>(); run_one<12>(); run_one<16>(); run_one<24>();
  return 0;
}
