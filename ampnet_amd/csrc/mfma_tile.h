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

// 16-byte load of a streamed (gathered, read-once-per-edge) tile chunk.  AMPCONV_NT_LOADS: non-temporal hint.
template <typename P>
__device__ __forceinline__ float4 stream_load4(const P *p) {
#ifdef AMPCONV_NT_LOADS
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4 *>(p);
#endif
}
#define STREAM_LOAD4(p) stream_load4(p)

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
// Bit 2 takes row bits 2 ^ 3 so that the four rows 4 sg + x (sg = 0..3) of a 4x4x1 phase-2 read land on four
// different chunk pairs (nt_accumulate: lanes (g, sg, rr) read row 4 sg + x, chunk 4 hf + g -- 2-way conflicts
// with bit 2 = row bit 2 alone); the other access patterns keep their cost (tools/lds_layout_check.py).
template <int DH>
__device__ __forceinline__ int swz(int j) {
  return (((((j >> 2) ^ (j >> 3)) & 1) << 2) | ((j >> 1) & 3)) & (DH / 4 - 1);
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
    if (valid) t.v[i] = STREAM_LOAD4(p);
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

// all-reduce over the 4 lane groups (lanes l, l^16, l^32, l^48): the 20 tokens of a C/D column.
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange whole 16- / 32-lane rows between two
// registers in the vector ALU (lane maps: tools/mfma4_probe.hip); `__shfl_xor` compiles to
// ds_bpermute_b32, an LDS-crossbar round trip the wave has to wait for (lgkmcnt).
// NB: take the builtin's result into an explicitly typed 2-vector and read .x / .y.  With `auto r` and
// r[0] / r[1], clang (ROCm 7.2) emitted extractvalue 0 for BOTH elements (a + b became 2 a).
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void swap16(float &a, float &b) {   // a' = [a0 b0 a2 b2], b' = [a1 b1 a3 b3] (16-lane rows)
  const u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap32(float &a, float &b) {   // a' = [a0 a1 b0 b1], b' = [a2 a3 b2 b3]
  const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x);
  b = __uint_as_float(r.y);
}
#ifdef AMPCONV_SHFL_BPERMUTE
__device__ __forceinline__ float groups_max(float x) {
  x = fmaxf(x, __shfl_xor(x, 16, 64));
  return fmaxf(x, __shfl_xor(x, 32, 64));
}
__device__ __forceinline__ float groups_sum(float x) {
  x += __shfl_xor(x, 16, 64);
  return x + __shfl_xor(x, 32, 64);
}
#else
__device__ __forceinline__ float groups_max(float x) {
  float a = x, b = x;
  swap16(a, b);                  // a = [x0 x0 x2 x2], b = [x1 x1 x3 x3]
  a = b = fmaxf(a, b);
  swap32(a, b);                  // a = [m01 m01 m01 m01], b = [m23 m23 m23 m23]
  return fmaxf(a, b);
}
__device__ __forceinline__ float groups_sum(float x) {
  float a = x, b = x;
  swap16(a, b);
  a = b = a + b;
  swap32(a, b);
  return a + b;
}
#endif

// ---- 4-granular products for the FIXED side's tail tokens 16..19 (v_mfma_f32_4x4x1_16b_f32).
// A 16x16x4 tile spends a whole 16-wide column tile on those four tokens (25 % useful); the 4x4x1
// form runs sixteen independent 4x4 outer products at the same per-MAC rate (9.3 vs 32 cycles per
// instruction, measured: tools/mfma4_probe.hip), so the tail columns cost a quarter.
//   lane l = (g = l >> 4, sg = (l >> 2) & 3, j = l & 3); block b = l >> 2:
//   A: lane holds A_b[i = l & 3]; B: B_b[j]; D: reg r of lane (b, j) = D_b[r][j] += A_b[r] B_b[j].
//   blgp 4 + x: every lane group reads its B from lanes 16 x .. 16 x + 15.
// Phase 1 (contraction over the channels): A = the ROW operand of the streamed 16-row tile as the
//   16x16x4 products use it (lane (m, ks) = tile[m][KK ks + kk]: block (ks, sg) = rows 4 sg .. 4 sg + 3,
//   channel KK ks + kk), B = tailop (lane (ks, sg, j) = fixed[16 + j][KK ks + kk]); D = the 16 x 4
//   result split over the four channel groups ks -> reduce_transpose sums them and leaves ONE
//   register: lane (g, sg, j) = result[row 4 sg + g][column 16 + j].
// Phase 2 (contraction over the streamed rows): that register is the B operand; with blgp 4 + x all
//   blocks see rows 4 sg + x, the A operand is image[4 sg + x][16 hf + 4 g + rr] (nt_accumulate) and
//   block (g, sg) accumulates out[16 hf + 4 g + rr][16 + j] over its rows; the four sg partial sums
//   are added once per unit (quads_sum).
#define MFMA4(a, b, c, blgp) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, (blgp))

__device__ __forceinline__ f32x4 mfma4_rows(float a, float b, f32x4 c, int x) {   // x is a constant after unrolling
  switch (x) {
    case 0: return MFMA4(a, b, c, 4);
    case 1: return MFMA4(a, b, c, 5);
    case 2: return MFMA4(a, b, c, 6);
    default: return MFMA4(a, b, c, 7);
  }
}

// fixed-side tail operand straight from global memory (once per unit), scaled; rows >= L read as zero
template <int DH>
__device__ __forceinline__ void tailop_from_global(float (&op)[TileCfg<DH>::KK], const float *base,
                                                   int64_t row_stride, float mul, int L, int lane) {
  using C = TileCfg<DH>;
  const int j = 16 + (lane & 3), ks = lane >> 4;
#pragma unroll
  for (int b = 0; b < C::KK / 4; ++b) {
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < L)
      x = *reinterpret_cast<const float4 *>(base + (int64_t)j * row_stride + C::KK * ks + 4 * b);
    op[4 * b + 0] = x.x * mul; op[4 * b + 1] = x.y * mul;
    op[4 * b + 2] = x.z * mul; op[4 * b + 3] = x.w * mul;
  }
}

// the same operand from a 20-row LDS image of the fixed tile (rows 16..19; every quad sg reads the same 16 bytes)
template <int DH>
__device__ __forceinline__ void tailop_from_lds(float (&op)[TileCfg<DH>::KK], const float *lds, int lane) {
  using C = TileCfg<DH>;
  const int j = 16 + (lane & 3), ks = lane >> 4;
#pragma unroll
  for (int b = 0; b < C::KK / 4; ++b) {
    const int chunk = (C::KK / 4) * ks + b;
    const float4 x = *reinterpret_cast<const float4 *>(lds + j * DH + ((chunk ^ swz<DH>(j)) << 2));
    op[4 * b + 0] = x.x; op[4 * b + 1] = x.y; op[4 * b + 2] = x.z; op[4 * b + 3] = x.w;
  }
}

// phase-1 partials (reg r of lane group ks = partial of row 4 sg + r) -> lane (g, sg, j) = row 4 sg + g
__device__ __forceinline__ float reduce_transpose(const f32x4 &x) {
  float x0 = x[0], x1 = x[1], x2 = x[2], x3 = x[3];
  swap16(x0, x1);                // x0 = [x0.g0 x1.g0 x0.g2 x1.g2], x1 = [x0.g1 x1.g1 x0.g3 x1.g3]
  float y01 = x0 + x1;           // group 0: x0 over ks 0,1; 1: x1 over ks 0,1; 2: x0 over ks 2,3; 3: x1 over ks 2,3
  swap16(x2, x3);
  float y23 = x2 + x3;
  swap32(y01, y23);              // y01 = [x0|01 x1|01 x2|01 x3|01], y23 = [x0|23 x1|23 x2|23 x3|23]
  return y01 + y23;
}

// phase 2 over one 16-row image: acc[hf] (block (g, sg), reg rr, lane j) += image[4 sg + x][16 hf + 4 g + rr] z[row 4 sg + x][j].
// `base` = this lane's float index of image[4 sg][4 g + rr] (lds_idx / tail_idx).  Address arithmetic kept out of the
// loop: the row step x is an ADD of x * DH (bits the base leaves zero: it folds into the ds_read offset); the XOR
// swizzle takes row bit 1 (= x >> 1) into chunk bit 0 and the channel half hf flips chunk bit 2.
//   PHYS = false: four variants of the address register per image, conflict-free reads.
//   PHYS = true : instruction p reads the PHYSICAL half p (one more immediate, two address variants): because the
//     swizzle flips chunk bit 2 for the rows of the quads with nt_flip(sg), such a lane's acc[p] belongs to the LOGICAL
//     half p ^ 1 and nt_fix_halves swaps the two accumulators of those lanes once per unit.  Chunk bit 2 then no longer
//     differs between the quads: every read is a 2-way bank conflict (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
//     0.00 -> 0.18) -- worth it only where the saved registers keep a wave per SIMD (the forward pass at 128).
__device__ __forceinline__ bool nt_flip(int lane) {            // chunk bit 2 of swz<32>(4 sg + x): row bits 2 ^ 3 = sg bits 0 ^ 1
  const int sg = (lane >> 2) & 3;
  return ((sg ^ (sg >> 1)) & 1) != 0;
}
template <int DH, bool PHYS>
__device__ __forceinline__ void nt_accumulate(f32x4 (&acc)[TileCfg<DH>::MC], const float *img, int base, float z) {
  const int b0 = (PHYS && DH == 32) ? (base & ~16) : base;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
#pragma unroll
    for (int hf = 0; hf < TileCfg<DH>::MC; ++hf) {
      const int idx = PHYS ? (b0 ^ ((x >> 1) << 2)) + x * DH + 16 * hf
                           : (base ^ (((x >> 1) << 2) | (hf << 4))) + x * DH;
      acc[hf] = mfma4_rows(img[idx], z, acc[hf], x);
    }
  }
}
template <int DH, bool PHYS>
__device__ __forceinline__ void nt_fix_halves(f32x4 (&acc)[TileCfg<DH>::MC], int lane) {
  if constexpr (PHYS && TileCfg<DH>::MC == 2) {
    const bool f = nt_flip(lane);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = acc[0][r], b = acc[1][r];
      acc[0][r] = f ? b : a;
      acc[1][r] = f ? a : b;
    }
  }
}

// sum over the four quads (sg) of every 16-lane row: row_ror:4, row_ror:8
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x);
__device__ __forceinline__ float quads_sum(float x);

// all-reduce over the 16 lanes of a DPP row (lane & 15): the columns of a C/D tile
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF,
                                                               0xF, true));
}
__device__ __forceinline__ float quads_sum(float x) {
  x += dpp_mov<0x124>(x);    // row_ror:4
  return x + dpp_mov<0x128>(x);    // row_ror:8
}
__device__ __forceinline__ float quads_max(float x) {
  x = fmaxf(x, dpp_mov<0x124>(x));
  return fmaxf(x, dpp_mov<0x128>(x));
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
