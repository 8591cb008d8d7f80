// Developer probe: cycles per MFMA, back to back on one wave, of the 16-bit 16x16 shapes (K = 32 vs the legacy K = 16).
//   ./tools/mfma_k_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(float *out, long long *cyc, int iters) {
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  bf16x8 a8, b8;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.01f + i); b8[i] = (__bf16)(1.f + i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c3, 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
  float *out; long long *cyc;
  (void)hipMalloc(&out, 1024); (void)hipHostMalloc(&cyc, 64);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    probe<0><<<1, 64>>>(out, cyc, iters);
    probe<1><<<1, 64>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  printf("16x16x32_bf16: %.2f clock ticks per MFMA; 16x16x16_bf16_1k: %.2f (s_memtime ticks, 100 MHz: compare the ratio)\n",
         cyc[0] / (4.0 * iters), cyc[1] / (4.0 * iters));
  return 0;
}
