// Shared pieces of the node-phase projection kernels (proj_gemm.hip: fp32 storage, proj_gemm_bf16.hip: bf16 storage).
#pragma once
#include "common.h"

namespace proj {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr int kFrag = 1024;          // one MFMA operand fragment: 64 lanes x 8 bf16
constexpr int kXcd = 8;

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned h) { return __builtin_bit_cast(float, h & 0xFFFF0000u); }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a, float b) {    // v_cvt_pk_f16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
#define MFMA32H(a, b, c)                                                                                         \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(proj::f16x8, a), __builtin_bit_cast(proj::f16x8, b), \
                                         (c), 0, 0, 0)

#define MFMA32(a, b, c)                                                                                          \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(proj::bf16x8, a), __builtin_bit_cast(proj::bf16x8, b), \
                                          (c), 0, 0, 0)

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform `lds_dst` + 16 * lane, source per lane.
// Inline assembly on purpose: behind the builtin hipcc orders every later LDS read after the DMA with
// `s_waitcnt vmcnt(0)` (the whole memory latency); the kernels wait for their DMAs themselves.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// two transposing 8-byte LDS reads = one 32x32x16 operand fragment whose 8 consecutive k per lane are ROWS of a
// row-major bf16 image (ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes, delivered column-major)
__device__ __forceinline__ i32x4 tr_frag(const char *p0, const char *p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
  const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
  return i32x4{ai[0], ai[1], bi[0], bi[1]};
}

inline int cu_count() {
  static const int n_cu = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    return n;
  }();
  return n_cu;
}

}  // namespace proj

// ---- bf16-storage entry points (proj_gemm_bf16.hip), dispatched by the C ABI in proj_gemm.hip
size_t ampconv_proj_weight_image_bytes_bf16(int N, int K);
bool ampconv_proj_supported_bf16(int N, int K);
int ampconv_proj_weight_images_bf16(int count, const ampconv_weight_image_t *jobs, hipStream_t stream);
int ampconv_proj_rows_bf16(const void *A, int64_t lda, int64_t M, int K, const void *wimage, int N, const void *bias,
                           const int32_t *rowptr, int L, void *out, int64_t ldc, const int32_t *nodes, int64_t n_nodes,
                           hipStream_t stream);
size_t ampconv_proj_wgrad_workspace_bytes_bf16(int64_t M, int Na, int Nb);
int ampconv_proj_wgrad_bf16(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t M, int Na, int Nb,
                            const int32_t *rowptr, int L, void *dW, void *colsum, void *workspace,
                            size_t workspace_bytes, const int32_t *nodes, int64_t n_nodes, hipStream_t stream);
