// MFMA edge-phase kernels for gfx950: L <= 20 tokens per node, dh in {16, 32}, fp32.
//
// One wavefront owns one (row, head) unit for the whole kernel: no workgroup barrier, no
// atomics, every output tile is written once by its owner.  Per edge the wave streams the
// source node's K and V head tiles (20 x dh floats each, whole 128-B / 64-B lines) through
// registers into a private, XOR-swizzled LDS image and cuts MFMA operands from it
// (mfma_tile.h).  All products run on v_mfma_f32_16x16x4_f32, i.e. exact fp32.
//
//   forward   S^T = K Q^T  (tokens of the source on MFMA rows, destination tokens on columns)
//             P^T = softmax over rows (in-register + 2 cross-lane steps)
//             O^T += V^T P^T           (P^T C/D registers are the B operand as they stand; the
//                                       sum over the in-edges of the destination accumulates
//                                       in the same MFMA accumulators = PyG's mean numerator)
// Reference arithmetic replaced: torch functional.py:6578 (scale), :6589 (QK^T), :6590
// (softmax), :6594 (PV) per edge, and the mean aggregation of amp_conv.py:11.
#include "mfma_tile.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kWavesPerBlock = 4;

struct FwdArgs {
  ampconv_view_t Q, K, V, O;
  const int32_t *rowptr, *col, *qidx;
  int64_t n_units;   // n_rows * H
  int L, H;
  float qscale;      // log2(e) / sqrt(dh)
};

// softmax over the 20 source tokens of one destination-token column held as
// (t0[0..3] = tokens 4g..4g+3, t1[0] = token 16+g) across the 4 lane groups g.
__device__ __forceinline__ void column_softmax(f32x4 &t0, f32x4 &t1, int L, int g) {
  if (L < kLmax) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * g + q >= L) t0[q] = -INFINITY;
    if (16 + g >= L) t1[0] = -INFINITY;
  }
  float m = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])), t1[0]);
  m = groups_max(m);
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] = fast_exp2(t0[q] - m);
  t1[0] = fast_exp2(t1[0] - m);
  float l = (t0[0] + t0[1]) + (t0[2] + t0[3]) + t1[0];
  l = groups_sum(l);
  const float inv = 1.f / l;
#pragma unroll
  for (int q = 0; q < 4; ++q) t0[q] *= inv;
  t1[0] *= inv;
}

template <int DH>
__global__ __launch_bounds__(64 * kWavesPerBlock) void fwd_mfma(FwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) float lds_all[kWavesPerBlock][2][C::TILE_FLOATS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  const int64_t r = unit / a.H;
  const int h = (int)(unit - r * a.H);
  const int L = a.L, g = lane >> 4;
  float *Kt = lds_all[wave][0], *Vt = lds_all[wave][1];

  const int beg = a.rowptr[r], end = a.rowptr[r + 1];
  const int64_t d = a.qidx ? a.qidx[r] : r;

  // fixed side: Q^T as the B operand (columns = destination tokens), pre-scaled so that
  // softmax(S) = exp2(S' - max S') / sum
  float qB[2][C::KK];
  {
    const float *qb = tile_ptr<const float>(a.Q, d, h);
    rowop_from_global<DH>(qB[0], qb, a.Q.row_stride, 0, true, a.qscale, L, lane);
    rowop_from_global<DH>(qB[1], qb, a.Q.row_stride, 1, true, a.qscale, L, lane);
  }
  if (L < kLmax) {      // token rows >= L of the images are read by the MFMAs: keep them finite
    tile_zero<DH>(Kt, lane);
    tile_zero<DH>(Vt, lane);
  }

  f32x4 OT[C::MC][2];
#pragma unroll
  for (int mc = 0; mc < C::MC; ++mc) OT[mc][0] = OT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  TileRegs<DH> kr, vr;
  if (beg < end) {
    const int64_t s = a.col[beg];
    tile_load<DH>(kr, tile_ptr<const float>(a.K, s, h), a.K.row_stride, L, lane);
    tile_load<DH>(vr, tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
  }
  for (int p = beg; p < end; ++p) {
    tile_to_lds<DH>(Kt, kr, 1.f, L, lane);
    tile_to_lds<DH>(Vt, vr, 1.f, L, lane);
    if (p + 1 < end) {                         // next edge's tiles fly while this one computes
      const int64_t s = a.col[p + 1];
      tile_load<DH>(kr, tile_ptr<const float>(a.K, s, h), a.K.row_stride, L, lane);
      tile_load<DH>(vr, tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
    }
    __builtin_amdgcn_wave_barrier();

    // S^T tiles [source-token tile mt][destination-token tile nt]
    f32x4 S[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      float kA[C::KK];
      rowop_from_lds<DH>(kA, Kt, mt, lane);
      S[mt][0] = S[mt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < C::KK; ++kk) {
        S[mt][0] = MFMA16(kA[kk], qB[0][kk], S[mt][0]);
        S[mt][1] = MFMA16(kA[kk], qB[1][kk], S[mt][1]);
      }
    }
    column_softmax(S[0][0], S[1][0], L, g);
    column_softmax(S[0][1], S[1][1], L, g);

    // O^T[channel tile mc][destination-token tile nt] += V^T P^T
#pragma unroll
    for (int mc = 0; mc < C::MC; ++mc) {
      float vA[5];
      colop_from_lds<DH>(vA, Vt, mc, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int s = 0; s < 4; ++s) OT[mc][nt] = MFMA16(vA[s], S[0][nt][s], OT[mc][nt]);
        OT[mc][nt] = MFMA16(vA[4], S[1][nt][0], OT[mc][nt]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // O^T C/D layout: lane (i' = lane & 15, g), reg q -> channel 4g + q + 16 mc, token i' + 16 nt
  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
  float *ob = tile_ptr<float>(a.O, r, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < C::MC; ++mc) {
        float4 o = make_float4(OT[mc][nt][0] * inv, OT[mc][nt][1] * inv, OT[mc][nt][2] * inv,
                               OT[mc][nt][3] * inv);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.O.row_stride + 4 * g + 16 * mc) = o;
      }
    }
  }
}

inline bool aligned16(const ampconv_view_t &v) {
  return ((uintptr_t)v.ptr % 16 == 0) && (v.node_stride % 4 == 0) && (v.row_stride % 4 == 0) &&
         (v.head_stride % 4 == 0);
}

}  // namespace

bool ampconv_mfma_supported(int L, int D, int H) {
  const int dh = D / H;
  return L >= 1 && L <= kLmax && (dh == 16 || dh == 32);
}

bool ampconv_mfma_views_ok(const ampconv_view_t *views, int n) {
  for (int i = 0; i < n; ++i)
    if (!aligned16(views[i])) return false;
  return true;
}

int ampconv_fwd_edge_mfma(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                          const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                          int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                          hipStream_t stream) {
  const int dh = D / H;
  FwdArgs a{Q, K, V, O, rowptr, col, qidx, n_rows * H, L, H, kLog2e / sqrtf((float)dh)};
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  if (dh == 32)
    fwd_mfma<32><<<(unsigned)blocks, 64 * kWavesPerBlock, 0, stream>>>(a);
  else
    fwd_mfma<16><<<(unsigned)blocks, 64 * kWavesPerBlock, 0, stream>>>(a);
  return ampconv_launch_status();
}
