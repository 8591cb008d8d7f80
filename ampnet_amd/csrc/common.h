// Shared helpers for the libampconv.so HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ampconv.h"

#define AMPCONV_WAVE 64

static inline int ampconv_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? AMPCONV_OK : (int)e;
}

static inline bool view_ok(const ampconv_view_t &v) { return v.ptr != nullptr; }

// element address of (node n, token l, head h) channel 0
template <typename T>
__device__ __forceinline__ T *tile_ptr(const ampconv_view_t &v, int64_t n, int h) {
  return reinterpret_cast<T *>(v.ptr) + n * v.node_stride + (int64_t)h * v.head_stride;
}

__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// ---- internal entry points (edge_generic.hip / edge_mfma.hip), dispatched by edge_api.hip
int ampconv_fwd_edge_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                             const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                             int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                             hipStream_t stream);
int ampconv_bwd_edge_dst_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                                 int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                                 hipStream_t stream);
int ampconv_bwd_edge_src_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                                 const float *cinv, int64_t n_src, int L, int D, int H,
                                 ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream);

// ---- MFMA fast path (edge_mfma.hip): L <= 20, dh in {16, 32}, 16-byte aligned views
bool ampconv_mfma_supported(int L, int D, int H);
bool ampconv_mfma_views_ok(const ampconv_view_t *views, int n);
int ampconv_fwd_edge_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                          const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                          int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                          hipStream_t stream);
int ampconv_bwd_edge_dst_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                              int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                              hipStream_t stream);
int ampconv_bwd_edge_src_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                              ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                              const float *cinv, int64_t n_src, int L, int D, int H,
                              ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream);

// ---- split-operand bf16 MFMA path (edge_mfma_split.hip): L <= 20, dh == 32; nprod = 9 or 6
bool ampconv_split_supported(int L, int D, int H);
int ampconv_fwd_edge_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                           const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                           int64_t n_rows, int L, int D, int H, ampconv_view_t O, hipStream_t stream);
int ampconv_bwd_edge_dst_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                               ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                               int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                               hipStream_t stream);
int ampconv_bwd_edge_src_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                               ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                               const float *cinv, int64_t n_src, int L, int D, int H,
                               ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream);
