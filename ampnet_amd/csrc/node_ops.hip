// Node-side helpers around the edge phase: PyG-style segment mean
// (`aggregate`), the in-degree mask of the out-projection, and the masked
// column sum that is out_proj.bias.grad.
#include "common.h"

namespace {

// out[n, :] = mean over the CSR segment of n of msg[eperm[p], :]   (0 if empty)
// reference: MessagePassing(aggr='mean'), amp_conv.py:11;
//            testing_message_passing_pyg.py:37-40 pins direction, mean and the zero rows.
__global__ void segment_mean_kernel(const float *__restrict__ msg,
                                    const int32_t *__restrict__ rowptr,
                                    const int32_t *__restrict__ eperm, int64_t F,
                                    float *__restrict__ out) {
  const int64_t n = blockIdx.x;
  const int beg = rowptr[n], end = rowptr[n + 1];
  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
  for (int64_t f = threadIdx.x; f < F; f += blockDim.x) {
    float acc = 0.f;
    for (int p = beg; p < end; ++p) acc += msg[(int64_t)eperm[p] * F + f];
    out[n * F + f] = acc * inv;
  }
}

// out[r, :] = scale_r * sum_{p in segment r} (w ? w[p] : 1) * rows[idx[p], :], one float4 column
// per thread, two edges in flight; the per-edge index and weight are wave-uniform (scalar loads)
__global__ __launch_bounds__(256) void gather_segment_sum_kernel(const float *__restrict__ rows,
                                                                  const int32_t *__restrict__ ptr,
                                                                  const int32_t *__restrict__ idx,
                                                                  const float *__restrict__ w, int mean,
                                                                  int64_t F, float *__restrict__ out) {
  const int64_t r = blockIdx.x;
  const int64_t f = ((int64_t)blockIdx.y * blockDim.x + threadIdx.x) * 4;
  if (f >= F) return;
  const int beg = ptr[r], end = ptr[r + 1];
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  int p = beg;
  for (; p + 1 < end; p += 2) {
    const float4 x = *reinterpret_cast<const float4 *>(rows + (int64_t)idx[p] * F + f);
    const float4 y = *reinterpret_cast<const float4 *>(rows + (int64_t)idx[p + 1] * F + f);
    const float wx = w ? w[p] : 1.f, wy = w ? w[p + 1] : 1.f;
    a.x = fmaf(wx, x.x, a.x); a.y = fmaf(wx, x.y, a.y); a.z = fmaf(wx, x.z, a.z); a.w = fmaf(wx, x.w, a.w);
    b.x = fmaf(wy, y.x, b.x); b.y = fmaf(wy, y.y, b.y); b.z = fmaf(wy, y.z, b.z); b.w = fmaf(wy, y.w, b.w);
  }
  if (p < end) {
    const float4 x = *reinterpret_cast<const float4 *>(rows + (int64_t)idx[p] * F + f);
    const float wx = w ? w[p] : 1.f;
    a.x = fmaf(wx, x.x, a.x); a.y = fmaf(wx, x.y, a.y); a.z = fmaf(wx, x.z, a.z); a.w = fmaf(wx, x.w, a.w);
  }
  const float sc = mean ? (end > beg ? 1.f / (float)(end - beg) : 0.f) : 1.f;
  *reinterpret_cast<float4 *>(out + r * F + f) =
      make_float4((a.x + b.x) * sc, (a.y + b.y) * sc, (a.z + b.z) * sc, (a.w + b.w) * sc);
}

// one wavefront per node (four per workgroup): 16-byte stores where the row allows them
template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(T *__restrict__ Y, const int32_t *__restrict__ rowptr, int64_t N,
                                                        int64_t F) {
  const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N || rowptr[n + 1] != rowptr[n]) return;
  const int lane = threadIdx.x & 63;
  T *row = Y + n * F;
  const int64_t bytes = F * (int64_t)sizeof(T);
  if ((((uintptr_t)Y | (uintptr_t)bytes) & 15) == 0) {
    for (int64_t i = 16 * lane; i < bytes; i += 1024)
      *reinterpret_cast<uint4 *>(reinterpret_cast<char *>(row) + i) = make_uint4(0u, 0u, 0u, 0u);
  } else {
    for (int64_t f = lane; f < F; f += 64) row[f] = (T)0.f;
  }
}

// partial[b, c] = sum over rows n in block b's slice with deg>0, tokens l of dY[n, l, c]
template <typename T>
__global__ void masked_colsum_partial(const T *__restrict__ dY,
                                      const int32_t *__restrict__ rowptr, int64_t N, int L, int D,
                                      int64_t rows_per_block, float *__restrict__ partial) {
  const int64_t n0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t n1 = n0 + rows_per_block < N ? n0 + rows_per_block : N;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // four independent chains: 4 loads in flight
    for (int64_t n = n0; n < n1; ++n) {
      if (rowptr[n + 1] == rowptr[n]) continue;
      const T *row = dY + n * (int64_t)L * D + c;
      int l = 0;
      for (; l + 4 <= L; l += 4) {
        a0 += (float)row[(int64_t)l * D];
        a1 += (float)row[(int64_t)(l + 1) * D];
        a2 += (float)row[(int64_t)(l + 2) * D];
        a3 += (float)row[(int64_t)(l + 3) * D];
      }
      for (; l < L; ++l) a0 += (float)row[(int64_t)l * D];
    }
    partial[(int64_t)blockIdx.x * D + c] = (a0 + a1) + (a2 + a3);
  }
}

__global__ void colsum_final(const float *__restrict__ partial, int nblocks, int D,
                             float *__restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= D) return;
  float acc = 0.f;
  for (int b = 0; b < nblocks; ++b) acc += partial[(int64_t)b * D + c];
  out[c] = acc;
}

}  // namespace

extern "C" int ampconv_segment_mean(const float *msg, const int32_t *rowptr, const int32_t *eperm,
                                    int64_t N, int64_t F, float *out, void *stream) {
  if (N < 0 || F < 0 || N > INT32_MAX) return AMPCONV_E_BADARG;
  if (N == 0 || F == 0) return AMPCONV_OK;
  if (!rowptr || !out) return AMPCONV_E_BADARG;
  segment_mean_kernel<<<(unsigned)N, 256, 0, (hipStream_t)stream>>>(msg, rowptr, eperm, F, out);
  return ampconv_launch_status();
}

extern "C" int ampconv_gather_segment_sum(const float *rows, const int32_t *ptr, const int32_t *idx,
                                          const float *w, int mean, int64_t N, int64_t F, float *out,
                                          void *stream) {
  if (N < 0 || F < 0 || N > INT32_MAX || F % 4 != 0 || F / 4 > (int64_t)65535 * 256) return AMPCONV_E_BADARG;
  if (N == 0 || F == 0) return AMPCONV_OK;
  if (!rows || !ptr || !idx || !out || (uintptr_t)rows % 16 || (uintptr_t)out % 16) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)N, (unsigned)((F / 4 + 255) / 256));
  gather_segment_sum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(rows, ptr, idx, w, mean, F, out);
  return ampconv_launch_status();
}

extern "C" int ampconv_mask_rows(void *Y, const int32_t *rowptr, int64_t N, int64_t F, int dtype,
                                 void *stream) {
  if (N < 0 || F < 0 || N > INT32_MAX) return AMPCONV_E_BADARG;
  if (N == 0 || F == 0) return AMPCONV_OK;
  if (!Y || !rowptr) return AMPCONV_E_BADARG;
  if (dtype == AMPCONV_BF16)
    mask_rows_kernel<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>((__bf16 *)Y, rowptr, N, F);
  else
    mask_rows_kernel<<<(unsigned)((N + 3) / 4), 256, 0, (hipStream_t)stream>>>((float *)Y, rowptr, N, F);
  return ampconv_launch_status();
}

// `out` must have room for D floats followed by a scratch area of
// AMPCONV_COLSUM_BLOCKS * D floats (see ampnet_amd/_lib.py); the two-step sum is
// deterministic (fixed block slices, fixed order).
extern "C" int ampconv_masked_colsum(const void *dY, const int32_t *rowptr, int64_t N, int L,
                                     int D, float *out, int dtype, void *stream) {
  if (N < 0 || L <= 0 || D <= 0) return AMPCONV_E_BADARG;
  if (!out) return AMPCONV_E_BADARG;
  const int nblocks = 1024;        // = COLSUM_BLOCKS of ampnet_amd/_lib.py
  float *partial = out + D;
  if (N == 0) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * D, (hipStream_t)stream);
    return e == hipSuccess ? AMPCONV_OK : (int)e;
  }
  if (!dY || !rowptr) return AMPCONV_E_BADARG;
  const int64_t rpb = (N + nblocks - 1) / nblocks;
  const int used = (int)((N + rpb - 1) / rpb);
  if (dtype == AMPCONV_BF16)
    masked_colsum_partial<<<used, 256, 0, (hipStream_t)stream>>>((const __bf16 *)dY, rowptr, N, L, D, rpb,
                                                               partial);
  else
    masked_colsum_partial<<<used, 256, 0, (hipStream_t)stream>>>((const float *)dY, rowptr, N, L, D, rpb,
                                                               partial);
  colsum_final<<<(D + 255) / 256, 256, 0, (hipStream_t)stream>>>(partial, used, D, out);
  return ampconv_launch_status();
}
