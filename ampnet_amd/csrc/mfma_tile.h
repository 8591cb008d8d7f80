// Building blocks of the MFMA edge kernels (edge_mfma.hip): one wavefront owns one
// (row, head) unit and a private LDS region; tiles are [L <= 20 tokens] x [DH channels].
//
// MFMA used: v_mfma_f32_16x16x4_f32 (exact fp32, guide cdna_hip_programming.md section 3):
//   A operand: lane l holds A[row = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][col = l & 15]
//   C/D      : lane l, reg q holds D[row = 4 * (l >> 4) + q][col = l & 15]
//
// Two operand shapes are cut from one LDS tile image (DESIGN.md "LDS tile image",
// tools/lds_layout_check.py for the bank analysis):
//   ROW operand   : lane (m = l & 15, ks = l >> 4) holds tile[row(m)][KK * ks + kk], kk < KK = DH/4
//                   (KK contiguous floats = ds_read_b128s).  The contraction index of the MFMA
//                   is k = 4 * kk + ks  <->  channel c = KK * ks + kk (any bijection works as
//                   long as both operands of a product use the same one).
//   COLUMN operand: lane (c' = l & 15, ks = l >> 4) holds tile[tok(step, ks)][c' + 16 * mc]
//                   with tok(step, ks) = 4 * ks + step (step < 4) and 16 + ks (step == 4):
//                   exactly the order in which a 16x16 C/D tile holds the 20 tokens when its
//                   second row tile uses the "quarter" map below, so a softmax result can be
//                   fed back as the other operand with no data movement.
// Token -> MFMA row map for the 20 tokens of one node (two 16-row tiles):
//   tile 0: row m          <-> token m
//   tile 1: row m, m % 4 == 0 <-> token 16 + m / 4   (other rows are don't-care)
//   In C/D layout tile 1 therefore has its 4 valid tokens in reg 0 of the 4 lane groups.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int kLmax = 20;   // tokens per node handled by the MFMA path

template <int DH>
struct TileCfg {
  static constexpr int CH = DH / 4;                       // 16-byte chunks per token row
  static constexpr int RPI = 64 / CH;                     // token rows per wave-wide load
  static constexpr int NLD = (kLmax + RPI - 1) / RPI;     // loads per tile (3 for DH=32, 2 for DH=16)
  static constexpr int KK = DH / 4;                       // k-steps over the channels
  static constexpr int MC = DH / 16;                      // 16-wide channel tiles
  static constexpr int TILE_FLOATS = kLmax * DH;
};

// XOR swizzle of the 16-byte chunk index inside a token row (conflict-free ds_write_b128,
// row-operand ds_read_b128 and column-operand ds_read_b32; see tools/lds_layout_check.py)
template <int DH>
__device__ __forceinline__ int swz(int j) {
  return ((((j >> 2) & 1) << 2) | ((j >> 1) & 3)) & (DH / 4 - 1);
}
template <int DH>
__device__ __forceinline__ int lds_idx(int j, int c) {
  return j * DH + ((((c >> 2) ^ swz<DH>(j)) << 2) | (c & 3));
}

// ---- global -> registers: lane (r = lane / CH, q = lane % CH) owns 16 B of NLD token rows
template <int DH>
struct TileRegs {
  float4 v[TileCfg<DH>::NLD];
};

template <int DH>
__device__ __forceinline__ void tile_load(TileRegs<DH> &t, const float *base, int64_t row_stride,
                                          int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < C::NLD; ++i) {
    const int j = r + C::RPI * i;
    if (j < L) t.v[i] = *reinterpret_cast<const float4 *>(base + (int64_t)j * row_stride + 4 * q);
  }
}

// registers -> swizzled LDS image, optionally scaled
template <int DH>
__device__ __forceinline__ void tile_to_lds(float *lds, const TileRegs<DH> &t, float mul, int L,
                                            int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < C::NLD; ++i) {
    const int j = r + C::RPI * i;
    if (j < L) {
      float4 x = t.v[i];
      x.x *= mul; x.y *= mul; x.z *= mul; x.w *= mul;
      *reinterpret_cast<float4 *>(lds + j * DH + ((q ^ swz<DH>(j)) << 2)) = x;
    }
  }
}

template <int DH>
__device__ __forceinline__ void tile_zero(float *lds, int lane) {
  for (int i = lane; i < TileCfg<DH>::TILE_FLOATS; i += AMPCONV_WAVE) lds[i] = 0.f;
}

// ---- two tiles (A then B, 2 x 20 token rows) streamed together: 40 rows = 5 full wave-wide
// loads at DH=32 (no predicated partial load), 3 at DH=16.  FULL = (L == 20): no row guards.
template <int DH>
struct PairRegs {
  static constexpr int NP = (2 * kLmax + TileCfg<DH>::RPI - 1) / TileCfg<DH>::RPI;
  float4 v[NP];
};

template <int DH, bool FULL>
__device__ __forceinline__ void pair_load(PairRegs<DH> &t, const float *baseA, int64_t strideA,
                                          const float *baseB, int64_t strideB, int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < PairRegs<DH>::NP; ++i) {
    const int R = r + C::RPI * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    const bool valid = (R < 2 * kLmax) && (FULL || j < L);
    // per-lane part as a 32-bit BYTE offset (loop invariant) on top of a tile base that is uniform
    // per edge: lets the compiler keep the base in SGPRs (global_load ... v_off, s[base:base+1])
    const unsigned boff = ((unsigned)j * (unsigned)(isB ? strideB : strideA) + 4u * (unsigned)q) * 4u;
    const char *p = reinterpret_cast<const char *>(isB ? baseB : baseA) + boff;
    if (valid) t.v[i] = *reinterpret_cast<const float4 *>(p);
  }
}

// registers -> the two swizzled LDS images (image B directly behind image A), scaled
template <int DH, bool FULL>
__device__ __forceinline__ void pair_to_lds(float *ldsA, const PairRegs<DH> &t, float mulA, float mulB,
                                            int L, int lane) {
  using C = TileCfg<DH>;
  const int r = lane / C::CH, q = lane % C::CH;
#pragma unroll
  for (int i = 0; i < PairRegs<DH>::NP; ++i) {
    const int R = r + C::RPI * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    const bool valid = (R < 2 * kLmax) && (FULL || j < L);
    if (valid) {
      const float mul = isB ? mulB : mulA;
      float4 x = t.v[i];
      x.x *= mul; x.y *= mul; x.z *= mul; x.w *= mul;
      *reinterpret_cast<float4 *>(ldsA + (isB ? C::TILE_FLOATS : 0) + j * DH + ((q ^ swz<DH>(j)) << 2)) = x;
    }
  }
}

// ---- neighbour indices of a segment, 64 at a time: one coalesced vector load per window and a
// v_readlane per edge instead of a dependent scalar memory load per edge (a scalar-cache miss
// on col[p] stalls the whole wave for ~2-3 k cycles before its tile loads can even be issued)
struct IdxWindow {
  int c0;     // first CSR/CSC position held by the window
  int idx;    // lane l holds idx[c0 + l]
  float w;    // lane l holds weight[c0 + l] (source pass: 1/in-degree of the edge's destination)
};
template <bool WEIGHTS>
__device__ __forceinline__ void idxwin_load(IdxWindow &win, const int32_t *idx, const float *wts, int c0,
                                            int end, int lane) {
  win.c0 = c0;
  const int q = c0 + lane < end ? c0 + lane : end - 1;
  win.idx = idx[q];
  if (WEIGHTS) win.w = wts[q];
}
template <bool WEIGHTS>
__device__ __forceinline__ int idxwin_get(IdxWindow &win, const int32_t *idx, const float *wts, int p,
                                          int end, int lane, float *w) {
  if (p - win.c0 >= AMPCONV_WAVE) idxwin_load<WEIGHTS>(win, idx, wts, p, end, lane);   // wave-uniform
  const int k = p - win.c0;
  if (WEIGHTS) *w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, win.w), k));
  return __builtin_amdgcn_readlane(win.idx, k);
}

// token held by MFMA row m of row-tile mt (quarter map for tile 1; always a valid row < 20)
__device__ __forceinline__ int row_token(int mt, int m) { return mt == 0 ? m : 16 + (m >> 2); }
// token held by (k-step, ks) of a column operand / by (reg, lane group) of a C/D tile pair
__device__ __forceinline__ int col_token(int step, int ks) { return step < 4 ? 4 * ks + step : 16 + ks; }

// ROW operand of row-tile mt from the LDS image: op[kk] = tile[row_token(mt, m)][KK*ks + kk]
template <int DH>
__device__ __forceinline__ void rowop_from_lds(float (&op)[TileCfg<DH>::KK], const float *lds, int mt,
                                               int lane) {
  using C = TileCfg<DH>;
  const int m = lane & 15, ks = lane >> 4;
  const int j = row_token(mt, m);
#pragma unroll
  for (int b = 0; b < C::KK / 4; ++b) {
    const int chunk = (C::KK / 4) * ks + b;
    const float4 x = *reinterpret_cast<const float4 *>(lds + j * DH + ((chunk ^ swz<DH>(j)) << 2));
    op[4 * b + 0] = x.x; op[4 * b + 1] = x.y; op[4 * b + 2] = x.z; op[4 * b + 3] = x.w;
  }
}

// ROW operand straight from global memory (used once per unit for the fixed side), scaled;
// token rows >= L read as zero.  `plain` = true uses token 16 + m for tile 1 (B-operand
// columns, no quarter map), false uses the quarter map (A-operand rows).
template <int DH>
__device__ __forceinline__ void rowop_from_global(float (&op)[TileCfg<DH>::KK], const float *base,
                                                  int64_t row_stride, int mt, bool plain, float mul,
                                                  int L, int lane) {
  using C = TileCfg<DH>;
  const int m = lane & 15, ks = lane >> 4;
  const int j = mt == 0 ? m : (plain ? 16 + m : 16 + (m >> 2));
#pragma unroll
  for (int b = 0; b < C::KK / 4; ++b) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < L)
      x = *reinterpret_cast<const float4 *>(base + (int64_t)j * row_stride + C::KK * ks + 4 * b);
    op[4 * b + 0] = x.x * mul; op[4 * b + 1] = x.y * mul;
    op[4 * b + 2] = x.z * mul; op[4 * b + 3] = x.w * mul;
  }
}

// COLUMN operand for channel tile mc: op[step] = tile[col_token(step, ks)][c' + 16*mc]
template <int DH>
__device__ __forceinline__ void colop_from_lds(float (&op)[5], const float *lds, int mc, int lane) {
  const int c = (lane & 15) + 16 * mc, ks = lane >> 4;
#pragma unroll
  for (int s = 0; s < 5; ++s) op[s] = lds[lds_idx<DH>(col_token(s, ks), c)];
}

// all-reduce over the 4 lane groups (lanes l, l^16, l^32, l^48): the 20 tokens of a C/D column
__device__ __forceinline__ float groups_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float groups_sum(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}

// all-reduce over the 16 lanes of a DPP row (lane & 15): the columns of a C/D tile
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF,
                                                               0xF, true));
}
__device__ __forceinline__ float row16_max(float x) {
  x = fmaxf(x, dpp_mov<0xB1>(x));    // quad_perm [1,0,3,2]
  x = fmaxf(x, dpp_mov<0x4E>(x));    // quad_perm [2,3,0,1]
  x = fmaxf(x, dpp_mov<0x141>(x));   // row_half_mirror
  return fmaxf(x, dpp_mov<0x140>(x));  // row_mirror
}
__device__ __forceinline__ float row16_sum(float x) {
  x += dpp_mov<0xB1>(x);
  x += dpp_mov<0x4E>(x);
  x += dpp_mov<0x141>(x);
  return x + dpp_mov<0x140>(x);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
constexpr float kNegBig = -1.0e30f;   // mask value: exp2(kNegBig - m) == 0, no inf/NaN arithmetic
