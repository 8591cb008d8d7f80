// Per-node products of the softmax-free AMPConv variant ("next" row 3 of SURVEY.md section 8f;
// host side: ampnet_amd/conv/linear.py).  Reference arithmetic: the softmax-free scaled-dot-product
// of src/ampnet/conv/custom_multihead_attn_forward.py:4173-4184, re-associated per node:
//   M_s = K_s^T V_s                    (outer,  per source node and head, dh x dh)
//   Obar_d = Q_d Mbar_d / sqrt(dh)     (apply,  Mbar = segment mean of M: ampconv_gather_segment_sum)
// and their transposes for the backward pass.  One wavefront per (node, head): the L x dh tiles are
// tiny (2.5 KB at L=20, dh=32), so these kernels are plain streaming passes over Q/K/V and M.
#include "common.h"

namespace {

struct LArgs {
  ampconv_view_t A, B;      // [node, token, head, channel] tiles
  float *M;                 // [N, H, dh, dh]
  int64_t n_units;
  int L, dh, H;
  float scale;
  int transpose;            // apply: use M^T
};

__device__ __forceinline__ void tile_to_lds(float *dst, const float *src, int L, int dh, int dhp, int64_t row_stride,
                                            int lane) {
  for (int idx = lane; idx < L * dh; idx += AMPCONV_WAVE) {
    const int l = idx / dh, c = idx - l * dh;
    dst[l * dhp + c] = src[(int64_t)l * row_stride + c];
  }
}

// M[u] = scale * A_u^T B_u
__global__ __launch_bounds__(AMPCONV_WAVE) void outer_kernel(LArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x, L = a.L, dh = a.dh, dhp = dh | 1;
  const int64_t u = blockIdx.x, n = u / a.H;
  const int h = (int)(u - n * a.H);
  float *As = lds, *Bs = lds + L * dhp;
  tile_to_lds(As, tile_ptr<const float>(a.A, n, h), L, dh, dhp, a.A.row_stride, lane);
  tile_to_lds(Bs, tile_ptr<const float>(a.B, n, h), L, dh, dhp, a.B.row_stride, lane);
  __syncthreads();
  float *out = a.M + u * (int64_t)dh * dh;
  for (int o = lane; o < dh * dh; o += AMPCONV_WAVE) {
    const int i = o / dh, j = o - i * dh;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) acc = fmaf(As[l * dhp + i], Bs[l * dhp + j], acc);
    out[o] = acc * a.scale;
  }
}

// Out_u = scale * A_u M_u   (or A_u M_u^T)
__global__ __launch_bounds__(AMPCONV_WAVE) void apply_kernel(LArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x, L = a.L, dh = a.dh, dhp = dh | 1;
  const int64_t u = blockIdx.x, n = u / a.H;
  const int h = (int)(u - n * a.H);
  float *As = lds, *Ms = lds + L * dhp;
  tile_to_lds(As, tile_ptr<const float>(a.A, n, h), L, dh, dhp, a.A.row_stride, lane);
  tile_to_lds(Ms, a.M + u * (int64_t)dh * dh, dh, dh, dhp, dh, lane);
  __syncthreads();
  float *ob = tile_ptr<float>(a.B, n, h);
  for (int o = lane; o < L * dh; o += AMPCONV_WAVE) {
    const int l = o / dh, c = o - l * dh;
    float acc = 0.f;
    if (a.transpose)
      for (int k = 0; k < dh; ++k) acc = fmaf(As[l * dhp + k], Ms[c * dhp + k], acc);
    else
      for (int k = 0; k < dh; ++k) acc = fmaf(As[l * dhp + k], Ms[k * dhp + c], acc);
    ob[(int64_t)l * a.B.row_stride + c] = acc * a.scale;
  }
}

int check(int64_t N, int L, int D, int H) {
  if (N < 0 || L <= 0 || D <= 0 || H <= 0 || D % H != 0) return AMPCONV_E_BADARG;
  if (N * H > INT32_MAX) return AMPCONV_E_BADARG;
  return AMPCONV_OK;
}

}  // namespace

extern "C" int ampconv_linear_outer(ampconv_view_t A, ampconv_view_t B, int64_t N, int L, int D, int H, float scale,
                                    float *M, void *stream) {
  if (int rc = check(N, L, D, H)) return rc;
  if (N == 0) return AMPCONV_OK;
  if (!view_ok(A) || !view_ok(B) || !M) return AMPCONV_E_BADARG;
  const int dh = D / H;
  LArgs a{A, B, M, N * H, L, dh, H, scale, 0};
  const size_t lds = (size_t)2 * L * (dh | 1) * sizeof(float);
  if (lds > 64 * 1024) return AMPCONV_E_BADARG;
  outer_kernel<<<(unsigned)a.n_units, AMPCONV_WAVE, lds, (hipStream_t)stream>>>(a);
  return ampconv_launch_status();
}

extern "C" int ampconv_linear_apply(ampconv_view_t A, const float *M, int transpose, int64_t N, int L, int D, int H,
                                    float scale, ampconv_view_t Out, void *stream) {
  if (int rc = check(N, L, D, H)) return rc;
  if (N == 0) return AMPCONV_OK;
  if (!view_ok(A) || !view_ok(Out) || !M) return AMPCONV_E_BADARG;
  const int dh = D / H;
  LArgs a{A, Out, const_cast<float *>(M), N * H, L, dh, H, scale, transpose};
  const size_t lds = (size_t)(L + dh) * (dh | 1) * sizeof(float);
  if (lds > 64 * 1024) return AMPCONV_E_BADARG;
  apply_kernel<<<(unsigned)a.n_units, AMPCONV_WAVE, lds, (hipStream_t)stream>>>(a);
  return ampconv_launch_status();
}
