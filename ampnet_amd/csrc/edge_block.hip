// MFMA edge-phase kernels for the shapes edge_mfma.hip does not take: L <= 64 tokens, even
// dh <= 64 (the reference's AMPGCN class defaults are L = 40, dh = 50: amp_gcn.py:23-26); fp32 arithmetic, fp32 or bf16
// storage (`bf16`: tiles are widened on the way into registers and rounded to nearest-even on the way out; long-segment
// partial tiles stay fp32).
//
// One WORKGROUP owns one (row, head) unit; its ntok = ceil(L / 16) wavefronts share the LDS
// images of the per-edge tiles and each owns one 16-token tile of the unit's own ("fixed") side:
//   forward / dst pass: wave w <-> destination tokens 16w .. 16w+15 (columns of S^T = K Q^T); the
//                       softmax runs along the source tokens = MFMA rows, inside the wave
//   src pass          : wave w <-> source tokens 16w .. 16w+15 (columns of S = Q K^T); the softmax
//                       would run across waves, so this pass takes P and delta from the statistics
//                       the destination pass stored (include/ampconv.h): no cross-wave reduction
// so no wave ever needs another wave's registers and every output tile has one owner (no atomics).
// Tiles are zero-padded in LDS to 16 ntok rows x DHP in {32, 64} channels; products run on
// v_mfma_f32_16x16x4_f32 (exact fp32) with the operand conventions of mfma_tile.h.
// Reference arithmetic replaced: the same lines as edge_mfma.hip (torch functional.py:6578-6594,
// amp_conv.py:11); backward per SURVEY.md A.2.
#include "mfma_tile.h"

namespace {

constexpr float kLog2eB = 1.4426950408889634f;
constexpr int kMaxTok = 4;       // 16-token tiles per node: L <= 64

template <int DHP>
__device__ __forceinline__ int bswz(int j) {
  return DHP == 64 ? (j & 15) : swz<32>(j);
}
template <int DHP>
__device__ __forceinline__ int bidx(int j, int c) {
  return j * DHP + ((((c >> 2) ^ bswz<DHP>(j)) << 2) | (c & 3));
}

// ---- storage type: fp32 or bf16 rows behind the same views (strides in elements)
struct Rows {              // base of one (node, head) tile
  const void *p;
  bool bf;
};
__device__ __forceinline__ Rows rows_of(const ampconv_view_t &v, int64_t n, int h, bool bf) {
  const int64_t off = n * v.node_stride + (int64_t)h * v.head_stride;
  return Rows{bf ? (const void *)(reinterpret_cast<const unsigned short *>(v.ptr) + off)
                 : (const void *)(reinterpret_cast<const float *>(v.ptr) + off), bf};
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xFFFF0000u); }
__device__ __forceinline__ float ld1(const Rows &r, int64_t off) {
  return r.bf ? bf_lo(reinterpret_cast<const unsigned short *>(r.p)[off]) : reinterpret_cast<const float *>(r.p)[off];
}
template <int VEC>
__device__ __forceinline__ void ldv(float (&out)[VEC], const Rows &r, int off) {
  if (r.bf) {
    const unsigned short *q = reinterpret_cast<const unsigned short *>(r.p) + off;
    if constexpr (VEC == 4) {
      const uint2 u = *reinterpret_cast<const uint2 *>(q);
      out[0] = bf_lo(u.x); out[1] = bf_hi(u.x); out[2] = bf_lo(u.y); out[3] = bf_hi(u.y);
    } else {
      const unsigned u = *reinterpret_cast<const unsigned *>(q);
      out[0] = bf_lo(u); out[1] = bf_hi(u);
    }
  } else {
    const float *q = reinterpret_cast<const float *>(r.p) + off;
    if constexpr (VEC == 4) {
      const float4 x = *reinterpret_cast<const float4 *>(q);
      out[0] = x.x; out[1] = x.y; out[2] = x.z; out[3] = x.w;
    } else {
      const float2 x = *reinterpret_cast<const float2 *>(q);
      out[0] = x.x; out[1] = x.y;
    }
  }
}
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {     // v_cvt_pk_bf16_f32, round to nearest even
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// ROW operand of token tile t from an LDS image: op[kk] = tile[16 t + m][KK ks + kk]
template <int DHP>
__device__ __forceinline__ void b_rowop_lds(float (&op)[DHP / 4], const float *lds, int t, int lane) {
  constexpr int KK = DHP / 4;
  const int m = lane & 15, ks = lane >> 4, j = 16 * t + m;
#pragma unroll
  for (int b = 0; b < KK / 4; ++b) {
    const int chunk = (KK / 4) * ks + b;
    const float4 x = *reinterpret_cast<const float4 *>(lds + j * DHP + ((chunk ^ bswz<DHP>(j)) << 2));
    op[4 * b + 0] = x.x; op[4 * b + 1] = x.y; op[4 * b + 2] = x.z; op[4 * b + 3] = x.w;
  }
}
// COLUMN operand of one token row: lane (m = lane & 15) takes the MC = DHP / 16 CONSECUTIVE channels MC m .. MC m + MC - 1
// (one ds_read_b128 / b64 instead of MC ds_read_b32 at a stride of 16 channels), i.e. MFMA row m of channel tile mc is
// channel MC m + mc.  A C/D tile [mc] then holds, in lane (n, g), register r: channel MC (4 g + r) + mc -- over (r, mc)
// the 4 MC consecutive channels from 4 MC g: store_ct writes them as whole vectors.
template <int DHP>
__device__ __forceinline__ void b_colop_lds(float (&op)[DHP / 16], const float *lds, int row, int lane) {
  const int m = lane & 15;
  if constexpr (DHP == 64) {
    const float4 x = *reinterpret_cast<const float4 *>(lds + row * DHP + ((m ^ bswz<DHP>(row)) << 2));
    op[0] = x.x; op[1] = x.y; op[2] = x.z; op[3] = x.w;
  } else {
    const float2 x = *reinterpret_cast<const float2 *>(lds + bidx<DHP>(row, 2 * m));
    op[0] = x.x; op[1] = x.y;
  }
}

// ROW operand of token tile t straight from global memory (the unit's fixed side), scaled;
// token rows >= L and channels >= dh read as zero
template <int DHP>
__device__ __forceinline__ void b_rowop_global(float (&op)[DHP / 4], const Rows &base, int64_t row_stride,
                                               int t, float mul, int L, int dh, int lane) {
  constexpr int KK = DHP / 4;
  const int m = lane & 15, ks = lane >> 4, j = 16 * t + m;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int c = KK * ks + kk;
    op[kk] = (j < L && c < dh) ? ld1(base, (int64_t)j * row_stride + c) * mul : 0.f;
  }
}

// ---- cooperative staging of two [L x dh] tiles (A then B) global -> registers -> LDS images.
// Thread (r0 = tid / DVP, cv = tid % DVP) owns vector column cv (VEC floats) of rows r0 + i RS.
template <int DHP, int VEC>
struct Stage {
  static constexpr int DVP = DHP / VEC;       // vector slots per padded row
  static constexpr int NP = DVP / 4;          // passes per tensor: 16 ntok / (64 ntok / DVP)
  float v[2][NP][VEC];
};

template <int DHP, int VEC>
__device__ __forceinline__ void stage_load(Stage<DHP, VEC> &s, const Rows &baseA, int64_t strideA,
                                           const Rows &baseB, int64_t strideB, int L, int dh, int tid,
                                           int nthreads) {
  using S = Stage<DHP, VEC>;
  const int cv = tid % S::DVP, r0 = tid / S::DVP, RS = nthreads / S::DVP, c = cv * VEC;
  // 32-bit element offsets from the (wave-uniform) tile base: one offset register per load instead of a 64-bit address
  // pair per pass and tensor (a tile spans L rows: far below 2^31 elements)
  const int sA = (int)strideA, sB = (int)strideB;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    const int j = r0 + i * RS;
    if (j < L && c < dh) {
      ldv<VEC>(s.v[0][i], baseA, j * sA + c);
      ldv<VEC>(s.v[1][i], baseB, j * sB + c);
    }
  }
}

template <int DHP, int VEC>
__device__ __forceinline__ void stage_store(float *ldsA, float *ldsB, const Stage<DHP, VEC> &s, float mulA,
                                            float mulB, int L, int dh, int tid, int nthreads) {
  using S = Stage<DHP, VEC>;
  const int cv = tid % S::DVP, r0 = tid / S::DVP, RS = nthreads / S::DVP, c = cv * VEC;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    const int j = r0 + i * RS;
    if (j < L && c < dh) {
      const int o = bidx<DHP>(j, c);
      if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(ldsA + o) = make_float4(s.v[0][i][0] * mulA, s.v[0][i][1] * mulA,
                                                             s.v[0][i][2] * mulA, s.v[0][i][3] * mulA);
        *reinterpret_cast<float4 *>(ldsB + o) = make_float4(s.v[1][i][0] * mulB, s.v[1][i][1] * mulB,
                                                             s.v[1][i][2] * mulB, s.v[1][i][3] * mulB);
      } else {
        *reinterpret_cast<float2 *>(ldsA + o) = make_float2(s.v[0][i][0] * mulA, s.v[0][i][1] * mulA);
        *reinterpret_cast<float2 *>(ldsB + o) = make_float2(s.v[1][i][0] * mulB, s.v[1][i][1] * mulB);
      }
    }
  }
}

// C/D tiles [channel tile mc][this wave's token tile] -> global rows of `v` (channels < dh, tokens < L); `bf`: bf16 rows.
// Lane (token n = lane & 15, g = lane >> 4), register r of tile mc = channel MC (4 g + r) + mc (b_colop_lds): per r the
// MC consecutive channels from MC (4 g + r) -- 4 MC consecutive channels per lane in all
template <int DHP, int VEC>
__device__ __forceinline__ void store_ct(const ampconv_view_t &v, int64_t node, int h, const f32x4 (&T)[DHP / 16],
                                         float scale, int tile, int L, int dh, int lane, bool bf) {
  constexpr int MC = DHP / 16;
  const int i = (lane & 15) + 16 * tile, g = lane >> 4;
  if (i >= L) return;
  const int64_t row = node * v.node_stride + (int64_t)h * v.head_stride + (int64_t)i * v.row_stride;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = MC * (4 * g + r);
    float x[MC];
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) x[mc] = T[mc][r] * scale;
    if (bf) {
      unsigned short *ob = reinterpret_cast<unsigned short *>(v.ptr) + row + c;
      if constexpr (MC == 4) {
        if constexpr (VEC == 4) {
          if (c < dh) *reinterpret_cast<uint2 *>(ob) = uint2{pk_bf16(x[0], x[1]), pk_bf16(x[2], x[3])};
        } else {
          if (c < dh) *reinterpret_cast<unsigned *>(ob) = pk_bf16(x[0], x[1]);
          if (c + 2 < dh) *reinterpret_cast<unsigned *>(ob + 2) = pk_bf16(x[2], x[3]);
        }
      } else {
        if (c < dh) *reinterpret_cast<unsigned *>(ob) = pk_bf16(x[0], x[1]);
      }
    } else {
      float *ob = reinterpret_cast<float *>(v.ptr) + row + c;
      if constexpr (MC == 4) {
        if constexpr (VEC == 4) {
          if (c < dh) *reinterpret_cast<float4 *>(ob) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
          if (c < dh) *reinterpret_cast<float2 *>(ob) = make_float2(x[0], x[1]);
          if (c + 2 < dh) *reinterpret_cast<float2 *>(ob + 2) = make_float2(x[2], x[3]);
        }
      } else {
        if (c < dh) *reinterpret_cast<float2 *>(ob) = make_float2(x[0], x[1]);
      }
    }
  }
}

struct BArgs {
  ampconv_view_t Q, K, V, dO, O, dK, dV;     // O = forward output / dQ
  const int32_t *ptr, *idx, *qidx;
  const float *cinv;
  const int32_t *spos;
  float *stats;
  HubArgs hub;              // long-segment plan (hub.hip): mode 1 skips long rows, mode 2 = one unit per chunk
  int64_t n_units;
  int L, dh, H, ntok;
  int bf16;                 // storage of Q / K / V / dO and of the outputs of the main pass (partial tiles: fp32)
  float qscale, oscale;
};

// softmax over the source tokens (MFMA rows of every token tile t < ntok) of one destination-token
// column; returns m + log2(sum).  S[t][q] = source token 16 t + 4 g + q.
template <int NT>
__device__ __forceinline__ float block_column_softmax(f32x4 (&S)[NT], int L, int g) {
  float m = kNegBig;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (16 * t + 4 * g + q >= L) S[t][q] = kNegBig;
      m = fmaxf(m, S[t][q]);
    }
  }
  m = groups_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      S[t][q] = fast_exp2(S[t][q] - m);
      l += S[t][q];
    }
  }
  l = groups_sum(l);
  const float inv = fast_rcp(l);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) S[t][q] *= inv;
  }
  return m + __builtin_amdgcn_logf(l);
}

// ---------------------------------------------------------------- forward
template <int DHP, int VEC, int NT>
__global__ __launch_bounds__(64 * NT) void fwd_block(BArgs a) {
  constexpr int KK = DHP / 4, MC = DHP / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nthreads = 64 * NT;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, blockIdx.x, a.n_units, a.H, r, onode, h, beg, end, deg)) return;    // block-uniform
  constexpr int ntok = NT;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  const bool bf = a.bf16 != 0;
  float *Kt = lds, *Vt = lds + 16 * ntok * DHP;
  const int64_t d = a.qidx ? a.qidx[r] : r;

  float qB[KK];
  b_rowop_global<DHP>(qB, rows_of(a.Q, d, h, bf), a.Q.row_stride, wave, a.qscale, L, dh, lane);
  for (int i = tid; i < 2 * 16 * ntok * DHP; i += nthreads) lds[i] = 0.f;     // padding stays zero
  f32x4 OT[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) OT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<DHP, VEC> st;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    stage_load<DHP, VEC>(st, rows_of(a.K, s, h, bf), a.K.row_stride,
                         rows_of(a.V, s, h, bf), a.V.row_stride, L, dh, tid, nthreads);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  __syncthreads();
  for (int p = beg; p < end; ++p) {
    stage_store<DHP, VEC>(Kt, Vt, st, 1.f, 1.f, L, dh, tid, nthreads);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float kA[KK];
      b_rowop_lds<DHP>(kA, Kt, t, lane);
      S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) S[t] = MFMA16(kA[kk], qB[kk], S[t]);
    }
    block_column_softmax<NT>(S, L, g);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float vA[MC];
        b_colop_lds<DHP>(vA, Vt, 16 * t + 4 * g + q, lane);
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) OT[mc] = MFMA16(vA[mc], S[t][q], OT[mc]);
      }
    }
    __syncthreads();
  }
  // hub pass: unnormalised partial tile, the combine pass applies 1/deg
  store_ct<DHP, VEC>(a.O, onode, h, OT, a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f), wave, L, dh,
                     lane, bf && a.hub.mode != 2);
}

// ---------------------------------------------------------------- backward, destination pass
#ifndef AMPCONV_BLOCK_BWD_WAVES      // developer A/B switch: minimum waves per SIMD of the two backward kernels
#define AMPCONV_BLOCK_BWD_WAVES 1
#endif
template <int DHP, int VEC, bool STATS, int NT>
__global__ __launch_bounds__(64 * NT, AMPCONV_BLOCK_BWD_WAVES) void bwd_dst_block(BArgs a) {
  constexpr int KK = DHP / 4, MC = DHP / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nthreads = 64 * NT;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, blockIdx.x, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  constexpr int ntok = NT;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  const bool bf = a.bf16 != 0;
  float *Kt = lds, *Vt = lds + 16 * ntok * DHP;
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN

  float qB[KK], gB[KK];
  b_rowop_global<DHP>(qB, rows_of(a.Q, r, h, bf), a.Q.row_stride, wave, a.qscale, L, dh, lane);
  b_rowop_global<DHP>(gB, rows_of(a.dO, r, h, bf), a.dO.row_stride, wave, inv, L, dh, lane);
  for (int i = tid; i < 2 * 16 * ntok * DHP; i += nthreads) lds[i] = 0.f;
  f32x4 dQT[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) dQT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<DHP, VEC> st;
  IdxWindow win;
  float pos_next = 0.f;                      // STATS: CSC position (int bits) in the window's weight slot
  const float *wts = reinterpret_cast<const float *>(a.spos);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, wts, p, end, lane, &pos_next);
    stage_load<DHP, VEC>(st, rows_of(a.K, s, h, bf), a.K.row_stride,
                         rows_of(a.V, s, h, bf), a.V.row_stride, L, dh, tid, nthreads);
  };
  if (beg < end) {
    idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
    fetch(beg);
  }
  __syncthreads();
  const int LS = 16 * ntok;
  for (int p = beg; p < end; ++p) {
    stage_store<DHP, VEC>(Kt, Vt, st, 1.f, 1.f, L, dh, tid, nthreads);
    float *sb = nullptr;
    if (STATS) sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos_next) * a.H + h) * (2 * LS);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT], dP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float kA[KK], vA[KK];
      b_rowop_lds<DHP>(kA, Kt, t, lane);
      b_rowop_lds<DHP>(vA, Vt, t, lane);
      S[t] = dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        S[t] = MFMA16(kA[kk], qB[kk], S[t]);
        dP[t] = MFMA16(vA[kk], gB[kk], dP[t]);
      }
    }
    const float lse = block_column_softmax<NT>(S, L, g);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[t][q], dP[t][q], part);
    }
    const float delta = groups_sum(part);
    if (STATS && g == 0) {                   // all LS columns: the source pass reads every one
      sb[(lane & 15) + 16 * wave] = lse;
      sb[LS + (lane & 15) + 16 * wave] = delta;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dS = S[t][q] * (dP[t][q] - delta);
        float kC[MC];
        b_colop_lds<DHP>(kC, Kt, 16 * t + 4 * g + q, lane);
#pragma unroll
        for (int mc = 0; mc < MC; ++mc) dQT[mc] = MFMA16(kC[mc], dS, dQT[mc]);
      }
    }
    __syncthreads();
  }
  store_ct<DHP, VEC>(a.O, onode, h, dQT, a.hub.mode == 2 ? 1.f : a.oscale, wave, L, dh, lane, bf && a.hub.mode != 2);
}

// ---------------------------------------------------------------- backward, source pass (needs the statistics)
template <int DHP, int VEC, int NT>
__global__ __launch_bounds__(64 * NT, AMPCONV_BLOCK_BWD_WAVES) void bwd_src_block(BArgs a) {
  constexpr int KK = DHP / 4, MC = DHP / 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nthreads = 64 * NT;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, blockIdx.x, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  constexpr int ntok = NT;
  const int L = a.L, dh = a.dh, g = lane >> 4, n = lane & 15;
  const bool bf = a.bf16 != 0;
  float *Qt = lds, *Gt = lds + 16 * ntok * DHP;

  float kB[KK], vB[KK];
  b_rowop_global<DHP>(kB, rows_of(a.K, s, h, bf), a.K.row_stride, wave, 1.f, L, dh, lane);
  b_rowop_global<DHP>(vB, rows_of(a.V, s, h, bf), a.V.row_stride, wave, 1.f, L, dh, lane);
  for (int i = tid; i < 2 * 16 * ntok * DHP; i += nthreads) lds[i] = 0.f;
  f32x4 dKT[MC], dVT[MC];
#pragma unroll
  for (int mc = 0; mc < MC; ++mc) dKT[mc] = dVT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stage<DHP, VEC> st;
  IdxWindow win;
  float inv_next = 0.f;
  // the edge's softmax statistics (2 LS floats, written by the destination pass at this CSC position) travel with its
  // tiles: one float per thread, requested an edge ahead and handed to the waves through LDS.  (Round 3 loaded them
  // from global memory inside the edge's own phase: half of the pass was that wait.)
  constexpr int LS = 16 * NT;
  float *sl = lds + 2 * 16 * NT * DHP;
  float stat_next = 0.f;
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv_next);
    if (tid < 2 * LS) stat_next = a.stats[((int64_t)p * a.H + h) * (2 * LS) + tid];
    stage_load<DHP, VEC>(st, rows_of(a.Q, d, h, bf), a.Q.row_stride,
                         rows_of(a.dO, d, h, bf), a.dO.row_stride, L, dh, tid, nthreads);
  };
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
    fetch(beg);
  }
  __syncthreads();
  const bool colok = n + 16 * wave < L;      // this lane's source token exists
  for (int p = beg; p < end; ++p) {
    stage_store<DHP, VEC>(Qt, Gt, st, a.qscale, inv_next, L, dh, tid, nthreads);
    if (tid < 2 * LS) sl[tid] = stat_next;
    const float *sb = sl;
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

#pragma unroll
    for (int t = 0; t < NT; ++t) {
      {                                      // destination tokens 16 t + 4 g + q
        const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sb + 16 * t + 4 * g);
        const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sb + LS + 16 * t + 4 * g);
        f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
        {
          float qA[KK], gA[KK];
          b_rowop_lds<DHP>(qA, Qt, t, lane);
          b_rowop_lds<DHP>(gA, Gt, t, lane);
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            S = MFMA16(qA[kk], kB[kk], S);
            dP = MFMA16(gA[kk], vB[kk], dP);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float pr = colok ? fast_exp2(S[q] - l4[q]) : 0.f;
          const float dS = pr * (dP[q] - d4[q]);
          float gC[MC], qC[MC];
          b_colop_lds<DHP>(gC, Gt, 16 * t + 4 * g + q, lane);
          b_colop_lds<DHP>(qC, Qt, 16 * t + 4 * g + q, lane);
#pragma unroll
          for (int mc = 0; mc < MC; ++mc) {
            dVT[mc] = MFMA16(gC[mc], pr, dVT[mc]);
            dKT[mc] = MFMA16(qC[mc], dS, dKT[mc]);
          }
        }
      }
    }
    __syncthreads();
  }
  store_ct<DHP, VEC>(a.dK, onode, h, dKT, a.hub.mode == 2 ? 1.f : a.oscale, wave, L, dh, lane, bf && a.hub.mode != 2);
  store_ct<DHP, VEC>(a.dV, onode, h, dVT, 1.f, wave, L, dh, lane, bf && a.hub.mode != 2);
}

inline int vec_of(const ampconv_view_t *views, int n, int dh, int esize) {
  int vec = dh % 4 == 0 ? 4 : 2;
  for (int i = 0; i < n; ++i) {
    const ampconv_view_t &v = views[i];
    while (vec > 1 && (((uintptr_t)v.ptr % (esize * vec)) || v.node_stride % vec || v.row_stride % vec ||
                       v.head_stride % vec))
      vec >>= 1;
  }
  return vec;
}

// kernel table: (dhp in {64, 32}) x (vec in {4, 2}) x (ntok 1..4)
typedef void (*BlockKernel)(BArgs);
template <template <int, int, int> class F>
struct KernelTable {
  template <int DHP, int VEC>
  static BlockKernel by_ntok(int ntok) {
    switch (ntok) {
      case 1: return F<DHP, VEC, 1>::get();
      case 2: return F<DHP, VEC, 2>::get();
      case 3: return F<DHP, VEC, 3>::get();
      default: return F<DHP, VEC, 4>::get();
    }
  }
  static BlockKernel get(int dhp, int vec, int ntok) {
    if (dhp == 64) return vec == 4 ? by_ntok<64, 4>(ntok) : by_ntok<64, 2>(ntok);
    return vec == 4 ? by_ntok<32, 4>(ntok) : by_ntok<32, 2>(ntok);
  }
};
template <int DHP, int VEC, int NT> struct FwdK { static BlockKernel get() { return fwd_block<DHP, VEC, NT>; } };
template <int DHP, int VEC, int NT> struct DstKS { static BlockKernel get() { return bwd_dst_block<DHP, VEC, true, NT>; } };
template <int DHP, int VEC, int NT> struct DstK { static BlockKernel get() { return bwd_dst_block<DHP, VEC, false, NT>; } };
template <int DHP, int VEC, int NT> struct SrcK { static BlockKernel get() { return bwd_src_block<DHP, VEC, NT>; } };

int launch_block(const BArgs &a, int dhp, BlockKernel k, hipStream_t stream, int extra_floats = 0) {
  if (a.n_units > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)a.n_units), block(64 * a.ntok);
  const size_t shmem = ((size_t)2 * 16 * a.ntok * dhp + extra_floats) * sizeof(float);
  hipLaunchKernelGGL(k, grid, block, shmem, stream, a);
  return ampconv_launch_status();
}

BArgs base_args(int64_t n_rows, int L, int D, int H, bool bf16) {
  BArgs a{};
  a.L = L; a.dh = D / H; a.H = H; a.ntok = (L + 15) / 16;
  a.bf16 = bf16;
  a.n_units = n_rows * H;
  a.qscale = kLog2eB / sqrtf((float)a.dh);
  return a;
}

}  // namespace

bool ampconv_block_supported(int L, int D, int H, const ampconv_view_t *views, int n, bool bf16) {
  const int dh = D / H;
  if (!(L >= 1 && L <= 16 * kMaxTok && dh >= 2 && dh <= 64 && dh % 2 == 0)) return false;
  return vec_of(views, n, dh, bf16 ? 2 : 4) >= 2;
}

int ampconv_block_stats_floats(int L) { return 2 * 16 * ((L + 15) / 16); }

int ampconv_fwd_edge_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                           const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                           ampconv_view_t O, HubArgs hub, bool bf16, hipStream_t stream) {
  BArgs a = base_args(n_rows, L, D, H, bf16);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.O = O;
  a.ptr = rowptr; a.idx = col; a.qidx = qidx;
  const ampconv_view_t views[] = {Q, K, V, O};
  const int vec = vec_of(views, 4, a.dh, bf16 ? 2 : 4), dhp = a.dh > 32 ? 64 : 32;
  if (ampconv_block_x3_supported(L, D, H, bf16))
    return ampconv_fwd_edge_block_x3(Q, K, V, rowptr, col, qidx, n_rows, L, D, H, O, hub, vec, bf16, stream);
  return launch_block(a, dhp, KernelTable<FwdK>::get(dhp, vec, a.ntok), stream);
}

int ampconv_bwd_edge_dst_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                               ampconv_view_t dQ, HubArgs hub, StatsArgs sa, bool bf16, hipStream_t stream) {
  BArgs a = base_args(n_rows, L, D, H, bf16);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.O = dQ;
  a.ptr = rowptr; a.idx = col; a.spos = sa.spos; a.stats = sa.stats;
  a.oscale = 1.f / sqrtf((float)a.dh);
  const ampconv_view_t views[] = {Q, K, V, dO, dQ};
  const int vec = vec_of(views, 5, a.dh, bf16 ? 2 : 4), dhp = a.dh > 32 ? 64 : 32;
  if (ampconv_block_x3_supported(L, D, H, bf16))
    return ampconv_bwd_edge_dst_block_x3(Q, K, V, dO, rowptr, col, n_rows, L, D, H, dQ, hub, sa, vec, bf16, stream);
  return launch_block(a, dhp, sa.stats ? KernelTable<DstKS>::get(dhp, vec, a.ntok) : KernelTable<DstK>::get(dhp, vec, a.ntok),
                      stream);
}

int ampconv_bwd_edge_src_block(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                               const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src,
                               int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub,
                               const float *stats, bool bf16, hipStream_t stream) {
  if (!stats) return AMPCONV_E_BADARG;
  BArgs a = base_args(n_src, L, D, H, bf16);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv; a.stats = const_cast<float *>(stats);
  a.oscale = 0.6931471805599453f;       // dK = ln2 * sum dS^T (Q * log2e / sqrt(dh))
  const ampconv_view_t views[] = {Q, K, V, dO, dK, dV};
  const int vec = vec_of(views, 6, a.dh, bf16 ? 2 : 4), dhp = a.dh > 32 ? 64 : 32;
  if (ampconv_block_x3_supported(L, D, H, bf16))
    return ampconv_bwd_edge_src_block_x3(Q, K, V, dO, cscptr, crow, cinv, n_src, L, D, H, dK, dV, hub, stats, vec, bf16, stream);
  return launch_block(a, dhp, KernelTable<SrcK>::get(dhp, vec, a.ntok), stream, 2 * 16 * a.ntok);
}
