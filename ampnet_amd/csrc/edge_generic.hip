// Shape-generic edge-phase kernels (any L, any dh): one wavefront per
// (row, head).  These cover every (L, D, H) the reference instantiates
// (dh = 3, 50, ...) and are the in-library cross-check for the
// MFMA fast path in edge_mfma.hip, which the dispatcher prefers when the shape
// fits it.  fp32 arithmetic, expf, fixed summation order (CSR order).
//
// Per edge e = (s -> d), head h   (reference: torch functional.py:6578-6594)
//   S = (Q[d,:,h] / sqrt(dh)) K[s,:,h]^T ; P = softmax_rows(S) ; O_e = P V[s,:,h]
// Mean over the in-edges of d        (reference: amp_conv.py:11, aggr='mean')
#include "common.h"

namespace {

struct FwdArgs {
  ampconv_view_t Q, K, V, O;
  const int32_t *rowptr, *col, *qidx;
  int L, dh, dhp, H;
  float scale;
};

struct BwdArgs {
  ampconv_view_t Q, K, V, dO, dQ, dK, dV;
  const int32_t *ptr;     // rowptr (dst pass) or cscptr (src pass)
  const int32_t *idx;     // col (dst pass) or crow (src pass)
  const float *cinv;      // src pass: 1/in-degree of the destination of each CSC edge
  int L, dh, dhp, H;
  float scale;
};

struct WArgs {
  ampconv_view_t Q, K;
  const int64_t *edge_index;
  int64_t E;
  float *W;
  int L, dh, dhp, H;
  float scale;
  int linear;     // 1: raw scaled scores (the reference's softmax-free variant), 0: softmax
};

__device__ __forceinline__ void load_tile(float *dst, const float *src, int L, int dh, int dhp,
                                          int64_t row_stride, float mul, int lane) {
  for (int idx = lane; idx < L * dh; idx += AMPCONV_WAVE) {
    int j = idx / dh, c = idx - j * dh;
    dst[j * dhp + c] = mul * src[(int64_t)j * row_stride + c];
  }
}

__device__ __forceinline__ void store_tile(float *dst, const float *src, int L, int dh, int dhp,
                                           int64_t row_stride, float mul, int lane) {
  for (int idx = lane; idx < L * dh; idx += AMPCONV_WAVE) {
    int j = idx / dh, c = idx - j * dh;
    dst[(int64_t)j * row_stride + c] = mul * src[j * dhp + c];
  }
}

__device__ __forceinline__ void zero_tile(float *dst, int n, int lane) {
  for (int idx = lane; idx < n; idx += AMPCONV_WAVE) dst[idx] = 0.f;
}

// P[j] = softmax_j( Qi . K[j] ) for one destination token (Qi already scaled).
__device__ __forceinline__ void softmax_row(const float *Qi, const float *Ks, float *P, int L,
                                            int dh, int dhp, int lane) {
  float m = -INFINITY;
  for (int j = lane; j < L; j += AMPCONV_WAVE) {
    float s = 0.f;
    for (int c = 0; c < dh; ++c) s = fmaf(Qi[c], Ks[j * dhp + c], s);
    P[j] = s;
    m = fmaxf(m, s);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int j = lane; j < L; j += AMPCONV_WAVE) {
    float p = expf(P[j] - m);
    P[j] = p;
    l += p;
  }
  l = wave_sum(l);
  float inv = 1.f / l;
  for (int j = lane; j < L; j += AMPCONV_WAVE) P[j] *= inv;
  __syncthreads();
}

__global__ __launch_bounds__(AMPCONV_WAVE) void fwd_generic(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t r = blockIdx.x / a.H;
  const int h = blockIdx.x - r * a.H;
  const int L = a.L, dh = a.dh, dhp = a.dhp, T = L * dhp;
  float *Qs = lds, *Ks = Qs + T, *Vs = Ks + T, *Os = Vs + T, *P = Os + T;
  const int beg = a.rowptr[r], end = a.rowptr[r + 1];
  const int64_t d = a.qidx ? a.qidx[r] : r;
  load_tile(Qs, tile_ptr<const float>(a.Q, d, h), L, dh, dhp, a.Q.row_stride, a.scale, lane);
  zero_tile(Os, T, lane);
  for (int p = beg; p < end; ++p) {
    const int64_t s = a.col[p];
    __syncthreads();
    load_tile(Ks, tile_ptr<const float>(a.K, s, h), L, dh, dhp, a.K.row_stride, 1.f, lane);
    load_tile(Vs, tile_ptr<const float>(a.V, s, h), L, dh, dhp, a.V.row_stride, 1.f, lane);
    __syncthreads();
    for (int i = 0; i < L; ++i) {
      softmax_row(Qs + i * dhp, Ks, P, L, dh, dhp, lane);
      for (int c = lane; c < dh; c += AMPCONV_WAVE) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(P[j], Vs[j * dhp + c], acc);
        Os[i * dhp + c] += acc;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
  store_tile(tile_ptr<float>(a.O, r, h), Os, L, dh, dhp, a.O.row_stride, inv, lane);
}

// dP, delta, dS for destination token i (P already in LDS); leaves dS[j] in LDS.
__device__ __forceinline__ void dsoftmax_row(const float *dOi, const float *Vs, const float *P,
                                             float *dS, int L, int dh, int dhp, int lane) {
  float part = 0.f;
  for (int j = lane; j < L; j += AMPCONV_WAVE) {
    float dp = 0.f;
    for (int c = 0; c < dh; ++c) dp = fmaf(dOi[c], Vs[j * dhp + c], dp);
    dS[j] = dp;
    part = fmaf(P[j], dp, part);
  }
  const float delta = wave_sum(part);
  for (int j = lane; j < L; j += AMPCONV_WAVE) dS[j] = P[j] * (dS[j] - delta);
  __syncthreads();
}

__global__ __launch_bounds__(AMPCONV_WAVE) void bwd_dst_generic(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t r = blockIdx.x / a.H;
  const int h = blockIdx.x - r * a.H;
  const int L = a.L, dh = a.dh, dhp = a.dhp, T = L * dhp;
  float *Qs = lds, *dOs = Qs + T, *Ks = dOs + T, *Vs = Ks + T, *dQs = Vs + T, *P = dQs + T,
        *dS = P + L;
  const int beg = a.ptr[r], end = a.ptr[r + 1];
  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
  load_tile(Qs, tile_ptr<const float>(a.Q, r, h), L, dh, dhp, a.Q.row_stride, a.scale, lane);
  load_tile(dOs, tile_ptr<const float>(a.dO, r, h), L, dh, dhp, a.dO.row_stride, inv, lane);
  zero_tile(dQs, T, lane);
  for (int p = beg; p < end; ++p) {
    const int64_t s = a.idx[p];
    __syncthreads();
    load_tile(Ks, tile_ptr<const float>(a.K, s, h), L, dh, dhp, a.K.row_stride, 1.f, lane);
    load_tile(Vs, tile_ptr<const float>(a.V, s, h), L, dh, dhp, a.V.row_stride, 1.f, lane);
    __syncthreads();
    for (int i = 0; i < L; ++i) {
      softmax_row(Qs + i * dhp, Ks, P, L, dh, dhp, lane);
      dsoftmax_row(dOs + i * dhp, Vs, P, dS, L, dh, dhp, lane);
      for (int c = lane; c < dh; c += AMPCONV_WAVE) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(dS[j], Ks[j * dhp + c], acc);
        dQs[i * dhp + c] += acc;
      }
      __syncthreads();
    }
  }
  __syncthreads();
  store_tile(tile_ptr<float>(a.dQ, r, h), dQs, L, dh, dhp, a.dQ.row_stride, a.scale, lane);
}

__global__ __launch_bounds__(AMPCONV_WAVE) void bwd_src_generic(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t s = blockIdx.x / a.H;
  const int h = blockIdx.x - s * a.H;
  const int L = a.L, dh = a.dh, dhp = a.dhp, T = L * dhp;
  float *Ks = lds, *Vs = Ks + T, *dKs = Vs + T, *dVs = dKs + T, *Qs = dVs + T, *dOs = Qs + T,
        *P = dOs + T, *dS = P + L;
  const int beg = a.ptr[s], end = a.ptr[s + 1];
  load_tile(Ks, tile_ptr<const float>(a.K, s, h), L, dh, dhp, a.K.row_stride, 1.f, lane);
  load_tile(Vs, tile_ptr<const float>(a.V, s, h), L, dh, dhp, a.V.row_stride, 1.f, lane);
  zero_tile(dKs, T, lane);
  zero_tile(dVs, T, lane);
  for (int p = beg; p < end; ++p) {
    const int64_t d = a.idx[p];
    const float inv = a.cinv[p];
    __syncthreads();
    load_tile(Qs, tile_ptr<const float>(a.Q, d, h), L, dh, dhp, a.Q.row_stride, a.scale, lane);
    load_tile(dOs, tile_ptr<const float>(a.dO, d, h), L, dh, dhp, a.dO.row_stride, inv, lane);
    __syncthreads();
    for (int i = 0; i < L; ++i) {
      softmax_row(Qs + i * dhp, Ks, P, L, dh, dhp, lane);
      dsoftmax_row(dOs + i * dhp, Vs, P, dS, L, dh, dhp, lane);
      for (int idx = lane; idx < L * dh; idx += AMPCONV_WAVE) {
        int j = idx / dh, c = idx - j * dh;
        dVs[j * dhp + c] = fmaf(P[j], dOs[i * dhp + c], dVs[j * dhp + c]);
        dKs[j * dhp + c] = fmaf(dS[j], Qs[i * dhp + c], dKs[j * dhp + c]);   // Qs carries the scale
      }
      __syncthreads();
    }
  }
  __syncthreads();
  store_tile(tile_ptr<float>(a.dK, s, h), dKs, L, dh, dhp, a.dK.row_stride, 1.f, lane);
  store_tile(tile_ptr<float>(a.dV, s, h), dVs, L, dh, dhp, a.dV.row_stride, 1.f, lane);
}

// attn_output_weights[e] = mean over heads of P (torch functional.py:6604-6606),
// one wavefront per edge in ORIGINAL edge order.
__global__ __launch_bounds__(AMPCONV_WAVE) void attn_weights_generic(WArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int64_t e = blockIdx.x;
  const int L = a.L, dh = a.dh, dhp = a.dhp, T = L * dhp;
  float *Qs = lds, *Ks = Qs + T, *P = Ks + T;
  const int64_t s = a.edge_index[e], d = a.edge_index[a.E + e];
  float *W = a.W + e * (int64_t)L * L;
  const float invH = 1.f / (float)a.H;
  for (int h = 0; h < a.H; ++h) {
    __syncthreads();
    load_tile(Qs, tile_ptr<const float>(a.Q, d, h), L, dh, dhp, a.Q.row_stride, a.scale, lane);
    load_tile(Ks, tile_ptr<const float>(a.K, s, h), L, dh, dhp, a.K.row_stride, 1.f, lane);
    __syncthreads();
    for (int i = 0; i < L; ++i) {
      if (a.linear) {
        for (int j = lane; j < L; j += AMPCONV_WAVE) {
          float sc = 0.f;
          for (int c = 0; c < dh; ++c) sc = fmaf(Qs[i * dhp + c], Ks[j * dhp + c], sc);
          P[j] = sc;
        }
        __syncthreads();
      } else {
        softmax_row(Qs + i * dhp, Ks, P, L, dh, dhp, lane);
      }
      for (int j = lane; j < L; j += AMPCONV_WAVE) {
        float w = P[j] * invH;
        if (h > 0) w += W[i * L + j];
        W[i * L + j] = w;
      }
      __syncthreads();
    }
  }
}

constexpr size_t kMaxLds = 160 * 1024;

template <typename K>
int set_lds(K kernel, size_t bytes) {
  if (bytes > kMaxLds) return AMPCONV_E_BADARG;
  if (bytes > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return (int)e;
  }
  return AMPCONV_OK;
}

inline int pad_odd(int dh) { return dh | 1; }

inline int check_shape(int L, int D, int H) {
  if (L <= 0 || D <= 0 || H <= 0 || D % H != 0) return AMPCONV_E_BADARG;
  return AMPCONV_OK;
}

}  // namespace

int ampconv_fwd_edge_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                             const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                             int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                             hipStream_t stream) {
  if (int rc = check_shape(L, D, H)) return rc;
  if (n_rows == 0) return AMPCONV_OK;
  if (n_rows * H > INT32_MAX) return AMPCONV_E_BADARG;
  FwdArgs a{Q, K, V, O, rowptr, col, qidx, L, D / H, pad_odd(D / H), H,
            1.f / sqrtf((float)(D / H))};
  size_t lds = ((size_t)4 * L * a.dhp + L) * sizeof(float);
  if (int rc = set_lds(fwd_generic, lds)) return rc;
  fwd_generic<<<(unsigned)(n_rows * H), AMPCONV_WAVE, lds, stream>>>(a);
  return ampconv_launch_status();
}

int ampconv_bwd_edge_dst_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                                 int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                                 hipStream_t stream) {
  if (int rc = check_shape(L, D, H)) return rc;
  if (n_rows == 0) return AMPCONV_OK;
  if (n_rows * H > INT32_MAX) return AMPCONV_E_BADARG;
  BwdArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dQ = dQ;
  a.ptr = rowptr; a.idx = col; a.cinv = nullptr;
  a.L = L; a.dh = D / H; a.dhp = pad_odd(a.dh); a.H = H;
  a.scale = 1.f / sqrtf((float)a.dh);
  size_t lds = ((size_t)5 * L * a.dhp + 2 * L) * sizeof(float);
  if (int rc = set_lds(bwd_dst_generic, lds)) return rc;
  bwd_dst_generic<<<(unsigned)(n_rows * H), AMPCONV_WAVE, lds, stream>>>(a);
  return ampconv_launch_status();
}

int ampconv_bwd_edge_src_generic(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                                 ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                                 const float *cinv, int64_t n_src, int L, int D, int H,
                                 ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream) {
  if (int rc = check_shape(L, D, H)) return rc;
  if (n_src == 0) return AMPCONV_OK;
  if (n_src * H > INT32_MAX) return AMPCONV_E_BADARG;
  BwdArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv;
  a.L = L; a.dh = D / H; a.dhp = pad_odd(a.dh); a.H = H;
  a.scale = 1.f / sqrtf((float)a.dh);
  size_t lds = ((size_t)6 * L * a.dhp + 2 * L) * sizeof(float);
  if (int rc = set_lds(bwd_src_generic, lds)) return rc;
  bwd_src_generic<<<(unsigned)(n_src * H), AMPCONV_WAVE, lds, stream>>>(a);
  return ampconv_launch_status();
}

static int attn_weights_impl(ampconv_view_t Q, ampconv_view_t K, const int64_t *edge_index, int64_t E, int L,
                             int D, int H, float *W, int dtype, int linear, void *stream);

extern "C" int ampconv_attn_weights(ampconv_view_t Q, ampconv_view_t K, const int64_t *edge_index,
                                    int64_t E, int L, int D, int H, float *W, int dtype,
                                    void *stream) {
  return attn_weights_impl(Q, K, edge_index, E, L, D, H, W, dtype, 0, stream);
}

extern "C" int ampconv_attn_scores(ampconv_view_t Q, ampconv_view_t K, const int64_t *edge_index,
                                   int64_t E, int L, int D, int H, float *W, int dtype, void *stream) {
  return attn_weights_impl(Q, K, edge_index, E, L, D, H, W, dtype, 1, stream);
}

static int attn_weights_impl(ampconv_view_t Q, ampconv_view_t K, const int64_t *edge_index, int64_t E, int L,
                             int D, int H, float *W, int dtype, int linear, void *stream) {
  if (dtype != AMPCONV_F32) return AMPCONV_E_DTYPE;
  if (int rc = check_shape(L, D, H)) return rc;
  if (E < 0 || E > INT32_MAX) return AMPCONV_E_BADARG;
  if (E == 0) return AMPCONV_OK;
  if (!view_ok(Q) || !view_ok(K) || !edge_index || !W) return AMPCONV_E_BADARG;
  WArgs a{Q, K, edge_index, E, W, L, D / H, pad_odd(D / H), H, 1.f / sqrtf((float)(D / H)), linear};
  size_t lds = ((size_t)2 * L * a.dhp + L) * sizeof(float);
  if (int rc = set_lds(attn_weights_generic, lds)) return rc;
  attn_weights_generic<<<(unsigned)E, AMPCONV_WAVE, lds, (hipStream_t)stream>>>(a);
  return ampconv_launch_status();
}
