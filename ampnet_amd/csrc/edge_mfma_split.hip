// Split-operand MFMA edge kernels for gfx950 (L <= 20, dh = 32, fp32 in / fp32 out).
//
// gfx950 runs fp32-input MFMA at 1/16 of the bf16 MFMA rate (MI355X_MICROARCH.md "Matrix
// cores"), which makes the exact-fp32 kernels of edge_mfma.hip MFMA-issue bound.  Here every
// fp32 operand element x is written EXACTLY as x = x1 + x2 + x3 with x1, x2, x3 in bf16 (3 x 8
// significant bits = the 24-bit fp32 significand; round-to-nearest splits, residuals are exact
// fp32 subtractions) and a product is evaluated as the sum of bf16 x bf16 partial products on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation:
//   NPROD = 9: all nine partial products -> every a*b is represented exactly before the fp32
//              accumulation (no operand rounding at all; error = fp32 accumulation only)
//   NPROD = 6: the three terms of order 2^-24 and below (x2*y3, x3*y2, x3*y3) are dropped
// Partial products are accumulated smallest first.
//
// Data path per edge (one wave per (row, head) unit, as in edge_mfma.hip): the two streamed
// 20 x 32 fp32 tiles are split ONCE, when they go from the load registers to LDS, into three
// bf16 plane images [20 tokens][32 channels] (64-byte rows, 16-byte chunks XOR-swizzled by
// kSwz[token >> 2]); both operand shapes are then read wide with no further arithmetic:
//   channel products (S, dP): A/B fragment = 8 consecutive channels of one token
//                             = one ds_read_b128 per plane
//   token products (PV, dQ, dK, dV): fragment = 8 k-slots of one channel; slots 0..3 hold tokens
//                             4g..4g+3, slots 4..7 hold tokens 16..19 in lane group g = 0 and
//                             zero-weighted elsewhere = two ds_read_b64_tr_b16 (hardware
//                             transpose read of a 4-token x 16-channel block) per plane
// The second 16-row tile of a score product holds tokens 16..19 in rows 0..3, so in C/D layout
// they sit in the 4 registers of lane group 0 -- exactly k-slots 4..7 of that group: softmax
// results become the B operand of the next product with no cross-lane movement.
#include "mfma_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kWavesPerBlock = 4;
constexpr int DH = 32;
constexpr int kPlaneBytes = kLmax * DH * 2;          // one bf16 plane of a tile: 1280 B
constexpr int kTileBytes = 3 * kPlaneBytes;          // 3840 B
typedef __attribute__((address_space(3))) char lds_char;

struct Frag3 {
  i32x4 p[3];   // three bf16 planes of an 8-element MFMA fragment
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32, RNE
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned h) { return __builtin_bit_cast(float, h & 0xFFFF0000u); }

// (x0, x1) -> packed bf16 pairs of the three planes; x == p1 + p2 + p3 exactly
struct Pair3 {
  int h1, h2, h3;
};
__device__ __forceinline__ Pair3 split_pair(float x0, float x1) {
  const unsigned a = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - lo_as_f32(a), r1 = x1 - hi_as_f32(a);
  const unsigned b = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - lo_as_f32(b), s1 = r1 - hi_as_f32(b);
  return Pair3{(int)a, (int)b, (int)cvt_pk_bf16(s0, s1)};
}

__device__ __forceinline__ void split8(const float (&x)[8], Frag3 &f) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const Pair3 q = split_pair(x[2 * t], x[2 * t + 1]);
    f.p[0][t] = q.h1; f.p[1][t] = q.h2; f.p[2][t] = q.h3;
  }
}
// C/D registers -> token-product fragment: k-slots 0..3 <- t0 (tokens 4g..4g+3), 4..7 <- t1
// (tokens 16..19, only meaningful in lane group 0; the caller zeroes t1 elsewhere)
__device__ __forceinline__ void split_cd(const f32x4 &t0, const f32x4 &t1, Frag3 &f) {
  const Pair3 a = split_pair(t0[0], t0[1]), b = split_pair(t0[2], t0[3]);
  const Pair3 c = split_pair(t1[0], t1[1]), d = split_pair(t1[2], t1[3]);
  f.p[0] = i32x4{a.h1, b.h1, c.h1, d.h1};
  f.p[1] = i32x4{a.h2, b.h2, c.h2, d.h2};
  f.p[2] = i32x4{a.h3, b.h3, c.h3, d.h3};
}

#define MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)

template <int NPROD>
__device__ __forceinline__ f32x4 mfma_split(const Frag3 &a, const Frag3 &b, f32x4 c) {
  if (NPROD == 9) {
    c = MFMA_BF16(a.p[2], b.p[2], c);
    c = MFMA_BF16(a.p[2], b.p[1], c);
    c = MFMA_BF16(a.p[1], b.p[2], c);
  }
  c = MFMA_BF16(a.p[2], b.p[0], c);
  c = MFMA_BF16(a.p[1], b.p[1], c);
  c = MFMA_BF16(a.p[0], b.p[2], c);
  c = MFMA_BF16(a.p[1], b.p[0], c);
  c = MFMA_BF16(a.p[0], b.p[1], c);
  c = MFMA_BF16(a.p[0], b.p[0], c);
  return c;
}

// ---- bf16 plane images in LDS ------------------------------------------------------------
// byte offset of 16-byte chunk `ch` (0..3) of token row `j` inside one plane
__device__ __forceinline__ int plane_off(int j, int ch) {
  // kSwz = {0, 3, 2, 1, 0}[j >> 2]: conflict-free ds_write_b64, ds_read_b128 and tr reads
  const int f = (4 - (j >> 2)) & 3;
  return j * 64 + ((ch ^ f) << 4);
}

// load registers (fp32, lane = token row r + 8 i, channels 4q..4q+3) -> split -> the three
// plane images of tile A (rows < 20) and tile B (rows >= 20), each 8 bytes per plane
template <bool FULL>
__device__ __forceinline__ void pair_to_planes(char *tileA, const PairRegs<DH> &t, float mulA, float mulB,
                                               int L, int lane) {
  const int r = lane >> 3, q = lane & 7;
#pragma unroll
  for (int i = 0; i < PairRegs<DH>::NP; ++i) {
    const int R = r + 8 * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    if (FULL || j < L) {
      const float mul = isB ? mulB : mulA;
      const float4 x = t.v[i];
      const Pair3 a = split_pair(x.x * mul, x.y * mul), b = split_pair(x.z * mul, x.w * mul);
      char *dst = tileA + (isB ? kTileBytes : 0) + plane_off(j, q >> 1) + ((q & 1) << 3);
      *reinterpret_cast<i32x2 *>(dst) = i32x2{a.h1, b.h1};
      *reinterpret_cast<i32x2 *>(dst + kPlaneBytes) = i32x2{a.h2, b.h2};
      *reinterpret_cast<i32x2 *>(dst + 2 * kPlaneBytes) = i32x2{a.h3, b.h3};
    }
  }
}

// channel-product fragment of row tile mt: lane (m = lane & 15, kg = lane >> 4) takes channels
// 8kg..8kg+7 of token m (tile 0) / token 16 + (m & 3) (tile 1: rows 0..3 valid, rest duplicates)
__device__ __forceinline__ void rowfrag(Frag3 &f, const char *tile, int mt, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = mt == 0 ? m : 16 + (m & 3);
  const char *p = tile + plane_off(j, kg);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) f.p[pl] = *reinterpret_cast<const i32x4 *>(p + pl * kPlaneBytes);
}

// token-product fragment of channel tile mc: lane (c' = lane & 15, kg = lane >> 4) takes, for
// channel c' + 16 mc, tokens 4kg..4kg+3 (slots 0..3) and tokens 16..19 (slots 4..7) through
// two transposed block reads; lane 4q + p of a 16-lane group addresses row q, channels 4p..4p+3
__device__ __forceinline__ void colfrag(Frag3 &f, const char *tile, int mc, int lane) {
  const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
  const int ch = 2 * mc + (pp >> 1), half = (pp & 1) << 3;
  const char *p0 = tile + plane_off(4 * kg + q, ch) + half;
  const char *p1 = tile + plane_off(16 + q, ch) + half;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p0 + pl * kPlaneBytes));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(p1 + pl * kPlaneBytes));
    const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
    f.p[pl] = i32x4{ai[0], ai[1], bi[0], bi[1]};
  }
}

__device__ __forceinline__ void planes_zero(char *tiles, int lane) {
  int *z = reinterpret_cast<int *>(tiles);
  for (int i = lane; i < 2 * kTileBytes / 4; i += AMPCONV_WAVE) z[i] = 0;
}

struct FwdArgs {
  ampconv_view_t Q, K, V, O;
  const int32_t *rowptr, *col, *qidx;
  int64_t n_units;
  int L, H;
  float qscale;
};

struct BwdArgs {
  ampconv_view_t Q, K, V, dO, dQ, dK, dV;
  const int32_t *ptr, *idx;
  const float *cinv;
  int64_t n_units;
  int L, H;
  float qscale, oscale;
};

// softmax over the 20 source tokens of one destination-token column: t0[q] = token 4g + q in
// every lane group g, t1[q] = token 16 + q in lane group 0 (t1 is forced to weight 0 elsewhere)
template <bool FULL>
__device__ __forceinline__ void column_softmax(f32x4 &t0, f32x4 &t1, int L, int g) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!FULL && 4 * g + q >= L) t0[q] = kNegBig;
    if (g != 0 || (!FULL && 16 + q >= L)) t1[q] = kNegBig;
  }
  float m = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])),
                  fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])));
  m = groups_max(m);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    t0[q] = fast_exp2(t0[q] - m);
    t1[q] = fast_exp2(t1[q] - m);
  }
  float l = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t1[0] + t1[1]) + (t1[2] + t1[3]));
  l = groups_sum(l);
  const float inv = fast_rcp(l);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    t0[q] *= inv;
    t1[q] *= inv;
  }
}

__device__ __forceinline__ void frag_from_global(Frag3 &f, const float *base, int64_t row_stride, int nt,
                                                 float mul, int L, int lane) {
  float x[8];
  rowop_from_global<DH>(x, base, row_stride, nt, true, mul, L, lane);
  split8(x, f);
}
// ---------------------------------------------------------------- forward
template <int NPROD, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void fwd_split(FwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  const int64_t r = unit / a.H;
  const int h = (int)(unit - r * a.H);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const int beg = a.rowptr[r], end = a.rowptr[r + 1];
  const int64_t d = a.qidx ? a.qidx[r] : r;

  Frag3 qB[2];
  {
    const float *qb = tile_ptr<const float>(a.Q, d, h);
    frag_from_global(qB[0], qb, a.Q.row_stride, 0, a.qscale, L, lane);
    frag_from_global(qB[1], qb, a.Q.row_stride, 1, a.qscale, L, lane);
  }
  if (!FULL) planes_zero(Kt, lane);
  f32x4 OT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc) OT[mc][0] = OT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegs<DH> kv;
  IdxWindow win;
  if (beg < end) {
    idxwin_load<false>(win, a.col, nullptr, beg, end, lane);
    const int64_t s = idxwin_get<false>(win, a.col, nullptr, beg, end, lane, nullptr);
    pair_load<DH, FULL>(kv, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_planes<FULL>(Kt, kv, 1.f, 1.f, L, lane);
    if (p + 1 < end) {
      const int64_t s = idxwin_get<false>(win, a.col, nullptr, p + 1, end, lane, nullptr);
      pair_load<DH, FULL>(kv, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                          tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
    }
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      Frag3 kA;
      rowfrag(kA, Kt, mt, lane);
      S[mt][0] = mfma_split<NPROD>(kA, qB[0], f32x4{0.f, 0.f, 0.f, 0.f});
      S[mt][1] = mfma_split<NPROD>(kA, qB[1], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    Frag3 pB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      column_softmax<FULL>(S[0][nt], S[1][nt], L, g);
      split_cd(S[0][nt], S[1][nt], pB[nt]);
    }
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      Frag3 vA;
      colfrag(vA, Vt, mc, lane);
      OT[mc][0] = mfma_split<NPROD>(vA, pB[0], OT[mc][0]);
      OT[mc][1] = mfma_split<NPROD>(vA, pB[1], OT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }

  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;
  float *ob = tile_ptr<float>(a.O, r, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < 2; ++mc) {
        float4 o = make_float4(OT[mc][nt][0] * inv, OT[mc][nt][1] * inv, OT[mc][nt][2] * inv,
                               OT[mc][nt][3] * inv);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.O.row_stride + 4 * g + 16 * mc) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- backward, destination pass
template <int NPROD, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void bwd_dst_split(BwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  const int64_t r = unit / a.H;
  const int h = (int)(unit - r * a.H);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const int beg = a.ptr[r], end = a.ptr[r + 1];
  const float inv = end > beg ? 1.f / (float)(end - beg) : 0.f;

  Frag3 qB[2], gB[2];
  {
    const float *qb = tile_ptr<const float>(a.Q, r, h);
    const float *gb = tile_ptr<const float>(a.dO, r, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      frag_from_global(qB[nt], qb, a.Q.row_stride, nt, a.qscale, L, lane);
      frag_from_global(gB[nt], gb, a.dO.row_stride, nt, inv, L, lane);
    }
  }
  if (!FULL) planes_zero(Kt, lane);
  f32x4 dQT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc) dQT[mc][0] = dQT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegs<DH> kv;
  IdxWindow win;
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, beg, end, lane, nullptr);
    pair_load<DH, FULL>(kv, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                        tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_planes<FULL>(Kt, kv, 1.f, 1.f, L, lane);
    if (p + 1 < end) {
      const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p + 1, end, lane, nullptr);
      pair_load<DH, FULL>(kv, tile_ptr<const float>(a.K, s, h), a.K.row_stride,
                          tile_ptr<const float>(a.V, s, h), a.V.row_stride, L, lane);
    }
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      Frag3 kA, vA;
      rowfrag(kA, Kt, mt, lane);
      S[mt][0] = mfma_split<NPROD>(kA, qB[0], f32x4{0.f, 0.f, 0.f, 0.f});
      S[mt][1] = mfma_split<NPROD>(kA, qB[1], f32x4{0.f, 0.f, 0.f, 0.f});
      rowfrag(vA, Vt, mt, lane);
      dP[mt][0] = mfma_split<NPROD>(vA, gB[0], f32x4{0.f, 0.f, 0.f, 0.f});
      dP[mt][1] = mfma_split<NPROD>(vA, gB[1], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    Frag3 sB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      column_softmax<FULL>(S[0][nt], S[1][nt], L, g);       // P^T; tile-1 weights are 0 for g != 0
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[0][nt][q], dP[0][nt][q], fmaf(S[1][nt][q], dP[1][nt][q], part));
      const float delta = groups_sum(part);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        S[0][nt][q] *= dP[0][nt][q] - delta;                 // dS^T
        S[1][nt][q] *= dP[1][nt][q] - delta;
      }
      split_cd(S[0][nt], S[1][nt], sB[nt]);
    }
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      Frag3 kC;
      colfrag(kC, Kt, mc, lane);
      dQT[mc][0] = mfma_split<NPROD>(kC, sB[0], dQT[mc][0]);
      dQT[mc][1] = mfma_split<NPROD>(kC, sB[1], dQT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }

  float *ob = tile_ptr<float>(a.dQ, r, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < 2; ++mc) {
        float4 o = make_float4(dQT[mc][nt][0] * a.oscale, dQT[mc][nt][1] * a.oscale,
                               dQT[mc][nt][2] * a.oscale, dQT[mc][nt][3] * a.oscale);
        *reinterpret_cast<float4 *>(ob + (int64_t)i * a.dQ.row_stride + 4 * g + 16 * mc) = o;
      }
    }
  }
}

// ---------------------------------------------------------------- backward, source pass
template <int NPROD, bool FULL>
__global__ __launch_bounds__(64 * kWavesPerBlock) void bwd_src_split(BwdArgs a) {
  using C = TileCfg<DH>;
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  const int64_t s = unit / a.H;
  const int h = (int)(unit - s * a.H);
  const int L = a.L, n = lane & 15;
  char *Qt = lds_all[wave], *Gt = Qt + kTileBytes;
  const int beg = a.ptr[s], end = a.ptr[s + 1];

  Frag3 kB[2], vB[2];
  {
    const float *kb = tile_ptr<const float>(a.K, s, h);
    const float *vb = tile_ptr<const float>(a.V, s, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      frag_from_global(kB[nt], kb, a.K.row_stride, nt, 1.f, L, lane);
      frag_from_global(vB[nt], vb, a.V.row_stride, nt, 1.f, L, lane);
    }
  }
  if (!FULL) planes_zero(Qt, lane);
  f32x4 dKT[2][2], dVT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc)
    dKT[mc][0] = dKT[mc][1] = dVT[mc][0] = dVT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegs<DH> qg;
  float inv = 0.f;
  IdxWindow win;
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, beg, end, lane, &inv);
    pair_load<DH, FULL>(qg, tile_ptr<const float>(a.Q, d, h), a.Q.row_stride,
                        tile_ptr<const float>(a.dO, d, h), a.dO.row_stride, L, lane);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_planes<FULL>(Qt, qg, a.qscale, inv, L, lane);
    if (p + 1 < end) {
      const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p + 1, end, lane, &inv);
      pair_load<DH, FULL>(qg, tile_ptr<const float>(a.Q, d, h), a.Q.row_stride,
                          tile_ptr<const float>(a.dO, d, h), a.dO.row_stride, L, lane);
    }
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      Frag3 qA, gA;
      rowfrag(qA, Qt, mt, lane);
      S[mt][0] = mfma_split<NPROD>(qA, kB[0], f32x4{0.f, 0.f, 0.f, 0.f});
      S[mt][1] = mfma_split<NPROD>(qA, kB[1], f32x4{0.f, 0.f, 0.f, 0.f});
      rowfrag(gA, Gt, mt, lane);
      dP[mt][0] = mfma_split<NPROD>(gA, vB[0], f32x4{0.f, 0.f, 0.f, 0.f});
      dP[mt][1] = mfma_split<NPROD>(gA, vB[1], f32x4{0.f, 0.f, 0.f, 0.f});
    }
    // row softmax over the source tokens (columns n, 16 + n across the 16 lanes of a DPP row);
    // rows: tile 0 reg q = destination token 4g + q, tile 1 reg q = token 16 + q in lane group 0
    const bool v0 = FULL || n < L, v1 = 16 + n < L;
    const bool g0 = lane < 16;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float s0 = v0 ? S[mt][0][q] : kNegBig, s1 = v1 ? S[mt][1][q] : kNegBig;
        const float m = row16_max(fmaxf(s0, s1));
        float p0 = fast_exp2(s0 - m), p1 = fast_exp2(s1 - m);
        float rinv = fast_rcp(row16_sum(p0 + p1));
        if (mt == 1 && !g0) rinv = 0.f;                    // rows of tile 1 exist in group 0 only
        p0 *= rinv;
        p1 *= rinv;
        const float delta = row16_sum(fmaf(p0, dP[mt][0][q], p1 * dP[mt][1][q]));
        S[mt][0][q] = p0;
        S[mt][1][q] = p1;
        dP[mt][0][q] = p0 * (dP[mt][0][q] - delta);
        dP[mt][1][q] = p1 * (dP[mt][1][q] - delta);
      }
    }
    Frag3 pB[2], sB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      split_cd(S[0][nt], S[1][nt], pB[nt]);
      split_cd(dP[0][nt], dP[1][nt], sB[nt]);
    }
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      Frag3 gC, qC;
      colfrag(gC, Gt, mc, lane);
      dVT[mc][0] = mfma_split<NPROD>(gC, pB[0], dVT[mc][0]);
      dVT[mc][1] = mfma_split<NPROD>(gC, pB[1], dVT[mc][1]);
      colfrag(qC, Qt, mc, lane);
      dKT[mc][0] = mfma_split<NPROD>(qC, sB[0], dKT[mc][0]);
      dKT[mc][1] = mfma_split<NPROD>(qC, sB[1], dKT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }

  const int g = lane >> 4;
  float *kb = tile_ptr<float>(a.dK, s, h), *vb = tile_ptr<float>(a.dV, s, h);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int j = n + 16 * nt;
    if (j < L) {
#pragma unroll
      for (int mc = 0; mc < 2; ++mc) {
        float4 k4 = make_float4(dKT[mc][nt][0] * a.oscale, dKT[mc][nt][1] * a.oscale,
                                dKT[mc][nt][2] * a.oscale, dKT[mc][nt][3] * a.oscale);
        float4 v4 = make_float4(dVT[mc][nt][0], dVT[mc][nt][1], dVT[mc][nt][2], dVT[mc][nt][3]);
        *reinterpret_cast<float4 *>(kb + (int64_t)j * a.dK.row_stride + 4 * g + 16 * mc) = k4;
        *reinterpret_cast<float4 *>(vb + (int64_t)j * a.dV.row_stride + 4 * g + 16 * mc) = v4;
      }
    }
  }
}

template <typename A, typename K9T, typename K9F, typename K6T, typename K6F>
int launch(const A &a, int nprod, bool full, K9T k9t, K9F k9f, K6T k6t, K6F k6f, hipStream_t stream) {
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  if (nprod == 9) {
    if (full) k9t<<<grid, block, 0, stream>>>(a); else k9f<<<grid, block, 0, stream>>>(a);
  } else {
    if (full) k6t<<<grid, block, 0, stream>>>(a); else k6f<<<grid, block, 0, stream>>>(a);
  }
  return ampconv_launch_status();
}

}  // namespace

bool ampconv_split_supported(int L, int D, int H) { return L >= 1 && L <= kLmax && D / H == DH; }

int ampconv_fwd_edge_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                           const int32_t *rowptr, const int32_t *col, const int32_t *qidx,
                           int64_t n_rows, int L, int D, int H, ampconv_view_t O, hipStream_t stream) {
  FwdArgs a{Q, K, V, O, rowptr, col, qidx, n_rows * H, L, H, kLog2e / sqrtf((float)DH)};
  return launch(a, nprod, L == kLmax, fwd_split<9, true>, fwd_split<9, false>, fwd_split<6, true>,
                fwd_split<6, false>, stream);
}

int ampconv_bwd_edge_dst_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                               ampconv_view_t dO, const int32_t *rowptr, const int32_t *col,
                               int64_t n_rows, int L, int D, int H, ampconv_view_t dQ,
                               hipStream_t stream) {
  BwdArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dQ = dQ;
  a.ptr = rowptr; a.idx = col; a.cinv = nullptr;
  a.n_units = n_rows * H; a.L = L; a.H = H;
  a.qscale = kLog2e / sqrtf((float)DH);
  a.oscale = 1.f / sqrtf((float)DH);
  return launch(a, nprod, L == kLmax, bwd_dst_split<9, true>, bwd_dst_split<9, false>,
                bwd_dst_split<6, true>, bwd_dst_split<6, false>, stream);
}

int ampconv_bwd_edge_src_split(int nprod, ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V,
                               ampconv_view_t dO, const int32_t *cscptr, const int32_t *crow,
                               const float *cinv, int64_t n_src, int L, int D, int H,
                               ampconv_view_t dK, ampconv_view_t dV, hipStream_t stream) {
  BwdArgs a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv;
  a.n_units = n_src * H; a.L = L; a.H = H;
  a.qscale = kLog2e / sqrtf((float)DH);
  a.oscale = 0.6931471805599453f;
  return launch(a, nprod, L == kLmax, bwd_src_split<9, true>, bwd_src_split<9, false>,
                bwd_src_split<6, true>, bwd_src_split<6, false>, stream);
}
