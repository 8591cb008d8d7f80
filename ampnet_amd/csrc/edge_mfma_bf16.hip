// bf16-storage edge kernels for gfx950 (BASELINE config 5): Q/K/V/O and the gradients live in HBM
// as bf16, every product runs on v_mfma_f32_16x16x32_bf16 with fp32 accumulation, softmax and all
// sums are fp32.  L <= 20, dh = 32 (or dh = 16 as a half-filled dh = 32 tile: `HALF`, see below).  Half the HBM bytes
// of the fp32 path and 1/16 of its MFMA
// cycles per product; the reference itself is fp32 only, so this is an extension checked against
// the fp32 oracle at bf16 tolerance (tests/test_gpu_parity.py::test_bf16_storage).
//
// Structure = edge_mfma_split.hip with ONE plane: a streamed 20 x 32 bf16 tile (64-byte token
// rows, 16 rows per wave-wide 16-B/lane load) goes straight into the swizzled LDS plane image;
// channel-product fragments are one ds_read_b128, token-product fragments two
// ds_read_b64_tr_b16 (hardware transpose), softmax results are converted to bf16 in registers
// and are the B operand of the next product as they stand (tokens 16..19 sit in lane group 0).
// Scales that the fp32 kernels fold into operands (log2e/sqrt(dh), 1/deg) are applied to the fp32
// accumulators instead.  Long segments (hubs) reduce into fp32 partial tiles (hub.hip).
#include <cstdlib>
#include "mfma_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kWavesPerBlock = 4;
constexpr int DH = 32;
// HALF (dh = 16, BASELINE config 3's head width): the head's 16 channels occupy the lower half of a 32-channel tile;
// the upper half is never loaded (zero operands: the 32-deep channel contraction is zero-padded), its output tile
// (mc = 1) is computed on zeros and not stored.  Functional, not tuned: half of every load and MFMA is padding.
constexpr int kRowBytes = DH * 2;                 // 64-byte token rows
constexpr int kTileBytes = kLmax * kRowBytes;     // 1280 B

#define MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)

__device__ __forceinline__ int cvt_pk_bf16(float a, float b) {
  f32x2 v = {a, b};
  return __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
}

// byte offset of 16-byte chunk `ch` (0..3) of token row `j` in the plane image (same swizzle as
// edge_mfma_split.hip: conflict-free ds_write_b128 / ds_read_b128 / transposed reads)
__device__ __forceinline__ int plane_off(int j, int ch) {
  const int f = (4 - (j >> 2)) & 3;
  return j * kRowBytes + ((ch ^ f) << 4);
}

// ---- streamed tile pair: lane (r = lane >> 2, q = lane & 3) owns 16 B (8 channels) of pair-rows
// r, r + 16, r + 32 (< 40); pair-rows 0..19 = tile A, 20..39 = tile B
struct PairRegsH {
  i32x4 v[3];
};

template <bool FULL, bool HALF>
__device__ __forceinline__ void pair_load_h(PairRegsH &t, const bf16_t *baseA, int64_t strideA,
                                            const bf16_t *baseB, int64_t strideB, int L, int lane) {
  const int r = lane >> 2, q = lane & 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int R = r + 16 * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    const bool valid = (R < 2 * kLmax) && (FULL || j < L) && (!HALF || q < 2);
    if (HALF) t.v[i] = i32x4{0, 0, 0, 0};
    const unsigned off = (unsigned)j * (unsigned)(isB ? strideB : strideA) + 8u * (unsigned)q;
    const bf16_t *p = (isB ? baseB : baseA) + off;
    if (valid) t.v[i] = *reinterpret_cast<const i32x4 *>(p);
  }
}

template <bool FULL>
__device__ __forceinline__ void pair_to_lds_h(char *tileA, const PairRegsH &t, int L, int lane) {
  const int r = lane >> 2, q = lane & 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int R = r + 16 * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    if ((R < 2 * kLmax) && (FULL || j < L))
      *reinterpret_cast<i32x4 *>(tileA + (isB ? kTileBytes : 0) + plane_off(j, q)) = t.v[i];
  }
}

__device__ __forceinline__ void tiles_zero(char *tiles, int lane) {
  int *z = reinterpret_cast<int *>(tiles);
  for (int i = lane; i < 2 * kTileBytes / 4; i += AMPCONV_WAVE) z[i] = 0;
}

// channel-product fragment of row tile mt from LDS.  Tile 1 uses the quarter map of mfma_tile.h:
// MFMA row m <-> token 16 + (m >> 2), so in C/D layout lane group g holds token 16 + g in reg 0
__device__ __forceinline__ i32x4 rowfrag(const char *tile, int mt, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = mt == 0 ? m : 16 + (m >> 2);
  return *reinterpret_cast<const i32x4 *>(tile + plane_off(j, kg));
}
// ... of column tile nt straight from global memory (fixed side; token rows >= L read as zero)
template <bool HALF>
__device__ __forceinline__ i32x4 rowfrag_global(const bf16_t *base, int64_t row_stride, int nt, int L,
                                                int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = nt == 0 ? m : 16 + m;
  i32x4 x = {0, 0, 0, 0};
  if (j < L && (!HALF || kg < 2)) x = *reinterpret_cast<const i32x4 *>(base + (int64_t)j * row_stride + 8 * kg);
  return x;
}
// token-product fragment of channel tile mc: k-slots 0..3 = tokens 4kg..4kg+3, slot 4 = token 16 + kg
// (slots 5..7 repeat it; the other operand holds zeros there)
__device__ __forceinline__ i32x4 colfrag(const char *tile, int mc, int lane) {
  const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
  const int ch = 2 * mc + (pp >> 1), half = (pp & 1) << 3;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + plane_off(4 * kg + q, ch) + half));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + plane_off(16 + kg, ch) + half));
  const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
  return i32x4{ai[0], ai[1], bi[0], bi[1]};
}
// Every transposed fragment of a product group is in its registers before the group's first MFMA issues, and no
// transposed read is scheduled in among the MFMAs: the interleaved schedule hipcc picks by itself -- an MFMA's operand
// register redefined by the next ds_read_b64_tr_b16 right behind it, consumed right behind the wait -- gave wrong sums in
// the weight-gradient kernel (csrc/proj_gemm.hip).  These kernels never showed it (thousands of bitwise
// identical repeated steps), at 4-6 waves per SIMD the wait costs nothing measurable, so they carry the fence too.
#ifdef AMPCONV_BF16_NOFENCE
#define TR_FRAG_FENCE() do {} while (0)
#else
#define TR_FRAG_FENCE()                                    \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
#endif
// C/D registers -> token-product fragment (t1[1..3] must already be zero: only reg 0 of tile 1 is a token)
__device__ __forceinline__ i32x4 cd_frag(const f32x4 &t0, const f32x4 &t1) {
  return i32x4{cvt_pk_bf16(t0[0], t0[1]), cvt_pk_bf16(t0[2], t0[3]), cvt_pk_bf16(t1[0], t1[1]),
               cvt_pk_bf16(t1[2], t1[3])};
}

// softmax over the 20 source tokens of one destination-token column; `sc` = log2e/sqrt(dh) is
// applied to the raw scores here.  t0[q] = token 4g + q, t1[0] = token 16 + g (regs 1..3 of tile 1 replicate it)
template <bool FULL>
__device__ __forceinline__ void column_softmax(f32x4 &t0, f32x4 &t1, float sc, int L, int g) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (!FULL && 4 * g + q >= L) t0[q] = kNegBig;
    if (q != 0 || (!FULL && 16 + g >= L)) t1[q] = kNegBig;
  }
  float m = fmaxf(fmaxf(fmaxf(t0[0], t0[1]), fmaxf(t0[2], t0[3])),
                  fmaxf(fmaxf(t1[0], t1[1]), fmaxf(t1[2], t1[3])));
  m = groups_max(m);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    t0[q] = fast_exp2((t0[q] - m) * sc);
    t1[q] = fast_exp2((t1[q] - m) * sc);
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

struct Args {
  ampconv_view_t Q, K, V, dO, O, dK, dV;   // O = forward output / dQ
  const int32_t *ptr, *idx, *qidx;
  const float *cinv;
  HubArgs hub;
  int64_t n_units;
  int L, H;
  float qscale, oscale;
};

// output tile store: C/D layout lane (i' = lane & 15, g), reg q -> channel 4g + q + 16 mc, token
// i' + 16 nt; `to_f32` = hub pass (fp32 partial tiles)
template <bool HALF>
__device__ __forceinline__ void store_tile(const ampconv_view_t &v, int64_t node, int h, const f32x4 (&T)[2][2],
                                           float scale, bool to_f32, int L, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int i = (lane & 15) + 16 * nt;
    if (i < L) {
#pragma unroll
      for (int mc = 0; mc < (HALF ? 1 : 2); ++mc) {
        const float a = T[mc][nt][0] * scale, b = T[mc][nt][1] * scale, c = T[mc][nt][2] * scale,
                    d = T[mc][nt][3] * scale;
        const int64_t off = node * v.node_stride + (int64_t)h * v.head_stride + (int64_t)i * v.row_stride +
                            4 * g + 16 * mc;
        if (to_f32)
          *reinterpret_cast<float4 *>(reinterpret_cast<float *>(v.ptr) + off) = make_float4(a, b, c, d);
        else
          *reinterpret_cast<i32x2 *>(reinterpret_cast<bf16_t *>(v.ptr) + off) =
              i32x2{cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)};
      }
    }
  }
}

// a unit without edges: its output rows are zero; nothing of its own side needs to be read (R-MAT graphs: half of the
// units of cfg5's main launch)
template <bool HALF>
__device__ __forceinline__ void store_zero_tile(const ampconv_view_t &v, int64_t node, int h, int L, int lane) {
  f32x4 Z[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc) Z[mc][0] = Z[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  store_tile<HALF>(v, node, h, Z, 0.f, false, L, lane);
}

// ---------------------------------------------------------------- forward
template <bool FULL, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock) void fwd_bf16(Args a) {
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) return store_zero_tile<HALF>(a.O, onode, h, a.L, lane);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const int64_t d = a.qidx ? a.qidx[r] : r;

  i32x4 qB[2];
  {
    const bf16_t *qb = tile_ptr<const bf16_t>(a.Q, d, h);
    qB[0] = rowfrag_global<HALF>(qb, a.Q.row_stride, 0, L, lane);
    qB[1] = rowfrag_global<HALF>(qb, a.Q.row_stride, 1, L, lane);
  }
  if (!FULL) tiles_zero(Kt, lane);
  f32x4 OT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc) OT[mc][0] = OT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsH kv;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    pair_load_h<FULL, HALF>(kv, tile_ptr<const bf16_t>(a.K, s, h), a.K.row_stride,
                      tile_ptr<const bf16_t>(a.V, s, h), a.V.row_stride, L, lane);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_lds_h<FULL>(Kt, kv, L, lane);
    if (p + 1 < end) fetch(p + 1);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const i32x4 kA = rowfrag(Kt, mt, lane);
      S[mt][0] = MFMA_BF16(kA, qB[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
      S[mt][1] = MFMA_BF16(kA, qB[1], (f32x4{0.f, 0.f, 0.f, 0.f}));
    }
    i32x4 pB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      column_softmax<FULL>(S[0][nt], S[1][nt], a.qscale, L, g);
      pB[nt] = cd_frag(S[0][nt], S[1][nt]);
    }
    i32x4 vA[2];
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) vA[mc] = colfrag(Vt, mc, lane);
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      OT[mc][0] = MFMA_BF16(vA[mc], pB[0], OT[mc][0]);
      OT[mc][1] = MFMA_BF16(vA[mc], pB[1], OT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const bool hubp = a.hub.mode == 2;
  store_tile<HALF>(a.O, onode, h, OT, hubp ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f), hubp, L, lane);
}

// ---------------------------------------------------------------- backward, destination pass
template <bool FULL, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock) void bwd_dst_bf16(Args a) {
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t r, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) return store_zero_tile<HALF>(a.O, onode, h, a.L, lane);
  const int L = a.L, g = lane >> 4;
  char *Kt = lds_all[wave], *Vt = Kt + kTileBytes;
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;     // dO is the gradient of the MEAN

  i32x4 qB[2], gB[2];
  {
    const bf16_t *qb = tile_ptr<const bf16_t>(a.Q, r, h);
    const bf16_t *gb = tile_ptr<const bf16_t>(a.dO, r, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      qB[nt] = rowfrag_global<HALF>(qb, a.Q.row_stride, nt, L, lane);
      gB[nt] = rowfrag_global<HALF>(gb, a.dO.row_stride, nt, L, lane);
    }
  }
  if (!FULL) tiles_zero(Kt, lane);
  f32x4 dQT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc) dQT[mc][0] = dQT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsH kv;
  IdxWindow win;
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    pair_load_h<FULL, HALF>(kv, tile_ptr<const bf16_t>(a.K, s, h), a.K.row_stride,
                      tile_ptr<const bf16_t>(a.V, s, h), a.V.row_stride, L, lane);
  };
  if (beg < end) {
    idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
    fetch(beg);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_lds_h<FULL>(Kt, kv, L, lane);
    if (p + 1 < end) fetch(p + 1);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const i32x4 kA = rowfrag(Kt, mt, lane), vA = rowfrag(Vt, mt, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        S[mt][nt] = MFMA_BF16(kA, qB[nt], (f32x4{0.f, 0.f, 0.f, 0.f}));
        dP[mt][nt] = MFMA_BF16(vA, gB[nt], (f32x4{0.f, 0.f, 0.f, 0.f}));
      }
    }
    i32x4 sB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      column_softmax<FULL>(S[0][nt], S[1][nt], a.qscale, L, g);   // P^T; tile-1 regs 1..3 get weight 0
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[0][nt][q], dP[0][nt][q], fmaf(S[1][nt][q], dP[1][nt][q], part));
      const float delta = groups_sum(part);
#pragma unroll
      for (int q = 0; q < 4; ++q) {                                // dS^T (1/deg applied here)
        S[0][nt][q] *= (dP[0][nt][q] - delta) * inv;
        S[1][nt][q] *= (dP[1][nt][q] - delta) * inv;
      }
      sB[nt] = cd_frag(S[0][nt], S[1][nt]);
    }
    i32x4 kC[2];
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) kC[mc] = colfrag(Kt, mc, lane);
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      dQT[mc][0] = MFMA_BF16(kC[mc], sB[0], dQT[mc][0]);
      dQT[mc][1] = MFMA_BF16(kC[mc], sB[1], dQT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const bool hubp = a.hub.mode == 2;
  store_tile<HALF>(a.O, onode, h, dQT, hubp ? 1.f : a.oscale, hubp, L, lane);
}

// ---------------------------------------------------------------- backward, source pass
template <bool FULL, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock) void bwd_src_bf16(Args a) {
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][2 * kTileBytes];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: unit, row, head and tile bases live in SGPRs
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) {
    store_zero_tile<HALF>(a.dK, onode, h, a.L, lane);
    return store_zero_tile<HALF>(a.dV, onode, h, a.L, lane);
  }
  const int L = a.L, n = lane & 15;
  char *Qt = lds_all[wave], *Gt = Qt + kTileBytes;

  i32x4 kB[2], vB[2];
  {
    const bf16_t *kb = tile_ptr<const bf16_t>(a.K, s, h);
    const bf16_t *vb = tile_ptr<const bf16_t>(a.V, s, h);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      kB[nt] = rowfrag_global<HALF>(kb, a.K.row_stride, nt, L, lane);
      vB[nt] = rowfrag_global<HALF>(vb, a.V.row_stride, nt, L, lane);
    }
  }
  if (!FULL) tiles_zero(Qt, lane);
  f32x4 dKT[2][2], dVT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc)
    dKT[mc][0] = dKT[mc][1] = dVT[mc][0] = dVT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsH qg;
  float inv = 0.f, inv_next = 0.f;
  IdxWindow win;
  auto fetch = [&](int p, float &w) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &w);
    pair_load_h<FULL, HALF>(qg, tile_ptr<const bf16_t>(a.Q, d, h), a.Q.row_stride,
                      tile_ptr<const bf16_t>(a.dO, d, h), a.dO.row_stride, L, lane);
  };
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
    fetch(beg, inv_next);
  }
  const bool v0 = FULL || n < L, v1 = 16 + n < L;
  for (int p = beg; p < end; ++p) {
    pair_to_lds_h<FULL>(Qt, qg, L, lane);
    inv = inv_next;
    if (p + 1 < end) fetch(p + 1, inv_next);
    __builtin_amdgcn_wave_barrier();

    f32x4 S[2][2], dP[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const i32x4 qA = rowfrag(Qt, mt, lane), gA = rowfrag(Gt, mt, lane);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        S[mt][nt] = MFMA_BF16(qA, kB[nt], (f32x4{0.f, 0.f, 0.f, 0.f}));
        dP[mt][nt] = MFMA_BF16(gA, vB[nt], (f32x4{0.f, 0.f, 0.f, 0.f}));
      }
    }
    // row softmax over the source tokens (columns n, 16 + n across the 16 lanes of a DPP row);
    // rows: tile 0 reg q = destination token 4g + q, tile 1 reg 0 = token 16 + g (quarter map).
    // After this block S holds P * (1/deg) (for dV) and dP holds dS (1/deg included).
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int q = 0; q < (mt == 0 ? 4 : 1); ++q) {
        const float s0 = v0 ? S[mt][0][q] : kNegBig, s1 = v1 ? S[mt][1][q] : kNegBig;
        const float m = row16_max(fmaxf(s0, s1));
        float p0 = fast_exp2((s0 - m) * a.qscale), p1 = fast_exp2((s1 - m) * a.qscale);
        const float rinv = fast_rcp(row16_sum(p0 + p1));
        p0 *= rinv;
        p1 *= rinv;
        const float delta = row16_sum(fmaf(p0, dP[mt][0][q], p1 * dP[mt][1][q]));
        S[mt][0][q] = p0 * inv;
        S[mt][1][q] = p1 * inv;
        dP[mt][0][q] = p0 * (dP[mt][0][q] - delta) * inv;
        dP[mt][1][q] = p1 * (dP[mt][1][q] - delta) * inv;
      }
    }
#pragma unroll
    for (int q = 1; q < 4; ++q)            // regs 1..3 of tile 1 are replicas, not tokens
      S[1][0][q] = S[1][1][q] = dP[1][0][q] = dP[1][1][q] = 0.f;
    i32x4 pB[2], sB[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      pB[nt] = cd_frag(S[0][nt], S[1][nt]);
      sB[nt] = cd_frag(dP[0][nt], dP[1][nt]);
    }
    i32x4 gC[2], qC[2];
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      gC[mc] = colfrag(Gt, mc, lane);
      qC[mc] = colfrag(Qt, mc, lane);
    }
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < 2; ++mc) {
      dVT[mc][0] = MFMA_BF16(gC[mc], pB[0], dVT[mc][0]);
      dVT[mc][1] = MFMA_BF16(gC[mc], pB[1], dVT[mc][1]);
      dKT[mc][0] = MFMA_BF16(qC[mc], sB[0], dKT[mc][0]);
      dKT[mc][1] = MFMA_BF16(qC[mc], sB[1], dKT[mc][1]);
    }
    __builtin_amdgcn_wave_barrier();
  }
  const bool hubp = a.hub.mode == 2;
  store_tile<HALF>(a.dK, onode, h, dKT, hubp ? 1.f : a.oscale, hubp, L, lane);
  store_tile<HALF>(a.dV, onode, h, dVT, 1.f, hubp, L, lane);
}

// ---------------------------------------------------------------- backward, source pass, transposing variant (round 5)
// The source pass needs two different axes of the 20 x 20 score tile: the softmax (and delta) sum over the SOURCE tokens
// j, the products dV = P^T dO, dK = dS^T Q sum over the DESTINATION tokens i.  bwd_src_bf16 keeps the tile as S[i][j]
// (i in the C/D registers: ready as the B operand of the products) and pays for the softmax with three 16-lane DPP
// reductions per row -- ~200 of its ~260 vector instructions per (edge, head), and the pass is bound by vector issue
// (profiles/r04_sq_counters.md: VALU active 0.82).  This variant computes the tile as S^T[j][i] -- own tokens on the MFMA
// rows, as the forward and destination passes do: the softmax is in-lane plus two cross-group swaps per column -- and
// turns P^T, dS^T round through a wave-private LDS image: each lane files its column as 8 + 2 bytes of row i, the B
// fragments come back by ds_read_b64_tr_b16 with the k index (i) contiguous, and the A fragments (dO^T, Q^T) are cut from
// 32-row images with the same contiguous k map.  Rows 20..31 of every image are zero (written once per unit).
constexpr int kTile32 = 32 * kRowBytes;       // 2 KiB: 32 token rows

template <bool FULL, bool HALF>
__device__ __forceinline__ void pair_to_lds_h32(char *tileA, const PairRegsH &t, int L, int lane) {
  const int r = lane >> 2, q = lane & 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int R = r + 16 * i;
    const bool isB = R >= kLmax;
    const int j = isB ? R - kLmax : R;
    if ((R < 2 * kLmax) && (FULL || j < L))
      *reinterpret_cast<i32x4 *>(tileA + (isB ? kTile32 : 0) + plane_off(j, q)) = t.v[i];
  }
}
// channel-product fragment with the PLAIN column map (column n of tile 1 = token 16 + n; rows 20..31 of the image are zero)
__device__ __forceinline__ i32x4 rowfrag_plain(const char *tile, int nt, int lane) {
  const int n = lane & 15, kg = lane >> 4;
  return *reinterpret_cast<const i32x4 *>(tile + plane_off(16 * nt + n, kg));
}
// ... of row tile mt straight from global memory with the quarter map (own side on the MFMA rows; once per unit)
template <bool HALF>
__device__ __forceinline__ i32x4 rowfrag_global_q(const bf16_t *base, int64_t row_stride, int mt, int L, int lane) {
  const int m = lane & 15, kg = lane >> 4;
  const int j = mt == 0 ? m : 16 + (m >> 2);
  i32x4 x = {0, 0, 0, 0};
  if (j < L && (!HALF || kg < 2)) x = *reinterpret_cast<const i32x4 *>(base + (int64_t)j * row_stride + 8 * kg);
  return x;
}
// token-product fragment with the k index contiguous: slots 0..7 of lane group kg = token rows 8 kg .. 8 kg + 7 of a
// 32-row image, 16-column block `blk` (channels of tile mc, or the source tokens of tile nt in a transposed image)
__device__ __forceinline__ i32x4 colfrag32(const char *tile, int blk, int lane) {
  const int q = (lane >> 2) & 3, pp = lane & 3, kg = lane >> 4;
  const int ch = 2 * blk + (pp >> 1), half = (pp & 1) << 3;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + plane_off(8 * kg + q, ch) + half));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4 *)(tile + plane_off(8 * kg + 4 + q, ch) + half));
  const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
  return i32x4{ai[0], ai[1], bi[0], bi[1]};
}
// a lane's column of a C/D tile pair (t0[q] = own token 4 g + q, t1_0 = own token 16 + g) -> row i of the transposed image
__device__ __forceinline__ void file_column(char *img, int i, const f32x4 &t0, float t1_0, int g) {
  *reinterpret_cast<i32x2 *>(img + plane_off(i, g >> 1) + 8 * (g & 1)) = i32x2{cvt_pk_bf16(t0[0], t0[1]), cvt_pk_bf16(t0[2], t0[3])};
  *reinterpret_cast<unsigned short *>(img + plane_off(i, 2) + 2 * g) = (unsigned short)cvt_pk_bf16(t1_0, 0.f);
}

template <bool FULL, bool HALF>
__global__ __launch_bounds__(64 * kWavesPerBlock) void bwd_src_bf16_tr(Args a) {
  __shared__ __attribute__((aligned(16))) char lds_all[kWavesPerBlock][4 * kTile32];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t unit = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (unit >= a.n_units) return;
  int64_t s, onode;
  int h, beg, end, deg;
  if (!map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  if (beg >= end && a.hub.mode != 2) {
    store_zero_tile<HALF>(a.dK, onode, h, a.L, lane);
    return store_zero_tile<HALF>(a.dV, onode, h, a.L, lane);
  }
  const int L = a.L, n = lane & 15, g = lane >> 4;
  char *Qt = lds_all[wave], *Gt = Qt + kTile32, *Pi = Gt + kTile32, *Si = Pi + kTile32;

  i32x4 kA[2], vA[2];
  {
    const bf16_t *kb = tile_ptr<const bf16_t>(a.K, s, h);
    const bf16_t *vb = tile_ptr<const bf16_t>(a.V, s, h);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      kA[mt] = rowfrag_global_q<HALF>(kb, a.K.row_stride, mt, L, lane);
      vA[mt] = rowfrag_global_q<HALF>(vb, a.V.row_stride, mt, L, lane);
    }
  }
  {   // all four images zero once: rows >= L (>= 20) of the streamed tiles and of the transposed images are never written
    int *z = reinterpret_cast<int *>(Qt);
    for (int i = lane; i < 4 * kTile32 / 4; i += AMPCONV_WAVE) z[i] = 0;
  }
  f32x4 dKT[2][2], dVT[2][2];
#pragma unroll
  for (int mc = 0; mc < 2; ++mc)
    dKT[mc][0] = dKT[mc][1] = dVT[mc][0] = dVT[mc][1] = f32x4{0.f, 0.f, 0.f, 0.f};

  PairRegsH qg;
  float inv = 0.f, inv_next = 0.f;
  IdxWindow win;
  auto fetch = [&](int p, float &w) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &w);
    pair_load_h<FULL, HALF>(qg, tile_ptr<const bf16_t>(a.Q, d, h), a.Q.row_stride,
                            tile_ptr<const bf16_t>(a.dO, d, h), a.dO.row_stride, L, lane);
  };
  if (beg < end) {
    idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
    fetch(beg, inv_next);
  }
  for (int p = beg; p < end; ++p) {
    pair_to_lds_h32<FULL, HALF>(Qt, qg, L, lane);
    inv = inv_next;
    if (p + 1 < end) fetch(p + 1, inv_next);
    __builtin_amdgcn_wave_barrier();

    // S^T = K Q^T and dP^T = V dO^T: own source tokens on the rows (quarter map), destination tokens on the columns
    f32x4 S[2][2], dP[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const i32x4 qB = rowfrag_plain(Qt, nt, lane), gB = rowfrag_plain(Gt, nt, lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        S[mt][nt] = MFMA_BF16(kA[mt], qB, (f32x4{0.f, 0.f, 0.f, 0.f}));
        dP[mt][nt] = MFMA_BF16(vA[mt], gB, (f32x4{0.f, 0.f, 0.f, 0.f}));
      }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      column_softmax<FULL>(S[0][nt], S[1][nt], a.qscale, L, g);        // P^T; tile-1 regs 1..3 get weight 0
      float part = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[0][nt][q], dP[0][nt][q], fmaf(S[1][nt][q], dP[1][nt][q], part));
      const float delta = groups_sum(part);
#pragma unroll
      for (int q = 0; q < 4; ++q) {                                     // dS^T and P^T, both with the edge's 1 / in-degree
        dP[0][nt][q] = S[0][nt][q] * (dP[0][nt][q] - delta) * inv;
        S[0][nt][q] *= inv;
      }
      dP[1][nt][0] = S[1][nt][0] * (dP[1][nt][0] - delta) * inv;
      S[1][nt][0] *= inv;
      // column i = 16 nt + n of the tile pair -> row i of the transposed images (tile 1: destination tokens 16..19 only)
      const int i = 16 * nt + n;
      if (i < kLmax && (FULL || i < L)) {
        file_column(Pi, i, S[0][nt], S[1][nt][0], g);
        file_column(Si, i, dP[0][nt], dP[1][nt][0], g);
      }
    }
    __builtin_amdgcn_wave_barrier();
    i32x4 pB[2], sB[2], gC[2], qC[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      pB[nt] = colfrag32(Pi, nt, lane);
      sB[nt] = colfrag32(Si, nt, lane);
    }
#pragma unroll
    for (int mc = 0; mc < (HALF ? 1 : 2); ++mc) {
      gC[mc] = colfrag32(Gt, mc, lane);
      qC[mc] = colfrag32(Qt, mc, lane);
    }
    TR_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < (HALF ? 1 : 2); ++mc) {
      dVT[mc][0] = MFMA_BF16(gC[mc], pB[0], dVT[mc][0]);
      dVT[mc][1] = MFMA_BF16(gC[mc], pB[1], dVT[mc][1]);
      dKT[mc][0] = MFMA_BF16(qC[mc], sB[0], dKT[mc][0]);
      dKT[mc][1] = MFMA_BF16(qC[mc], sB[1], dKT[mc][1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 7" ::: "memory");       // (the next edge's reads do not redefine fragment registers right behind these MFMAs)
    __builtin_amdgcn_wave_barrier();
  }
  const bool hubp = a.hub.mode == 2;
  store_tile<HALF>(a.dK, onode, h, dKT, hubp ? 1.f : a.oscale, hubp, L, lane);
  store_tile<HALF>(a.dV, onode, h, dVT, 1.f, hubp, L, lane);
}

// kernels[2 * half + full]
typedef void (*EdgeKernel)(Args);
int launch(Args &a, int L, int D, int H, EdgeKernel const (&kernels)[4], hipStream_t stream) {
  const int dh = D / H;
  if (dh != DH && dh != DH / 2) return AMPCONV_E_BADARG;
  a.qscale = kLog2e / sqrtf((float)dh);
  a.oscale = 1.f / sqrtf((float)dh);
  const int64_t blocks = (a.n_units + kWavesPerBlock - 1) / kWavesPerBlock;
  if (blocks > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  kernels[2 * (dh == DH / 2) + (L == kLmax)]<<<grid, block, 0, stream>>>(a);
  return ampconv_launch_status();
}

inline bool aligned16h(const ampconv_view_t &v) {
  return ((uintptr_t)v.ptr % 16 == 0) && (v.node_stride % 8 == 0) && (v.row_stride % 8 == 0) &&
         (v.head_stride % 8 == 0);
}

}  // namespace

bool ampconv_bf16_supported(int L, int D, int H, const ampconv_view_t *views, int n) {
  if (!(L >= 1 && L <= kLmax && (D / H == DH || D / H == DH / 2) && D % H == 0)) return false;
  for (int i = 0; i < n; ++i)
    if (!aligned16h(views[i])) return false;
  return true;
}

int ampconv_fwd_edge_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                          const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                          ampconv_view_t O, HubArgs hub, hipStream_t stream) {
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.O = O;
  a.ptr = rowptr; a.idx = col; a.qidx = qidx; a.hub = hub;
  a.n_units = n_rows * H; a.L = L; a.H = H;
  static const EdgeKernel kernels[4] = {fwd_bf16<false, false>, fwd_bf16<true, false>, fwd_bf16<false, true>, fwd_bf16<true, true>};
  return launch(a, L, D, H, kernels, stream);
}

int ampconv_bwd_edge_dst_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                              const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D,
                              int H, ampconv_view_t dQ, HubArgs hub, hipStream_t stream) {
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.O = dQ;
  a.ptr = rowptr; a.idx = col; a.hub = hub;
  a.n_units = n_rows * H; a.L = L; a.H = H;
  static const EdgeKernel kernels[4] = {bwd_dst_bf16<false, false>, bwd_dst_bf16<true, false>, bwd_dst_bf16<false, true>, bwd_dst_bf16<true, true>};
  return launch(a, L, D, H, kernels, stream);
}

int ampconv_bwd_edge_src_bf16(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                              const int32_t *cscptr, const int32_t *crow, const float *cinv,
                              int64_t n_src, int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV,
                              HubArgs hub, hipStream_t stream) {
  Args a{};
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv; a.hub = hub;
  a.n_units = n_src * H; a.L = L; a.H = H;
  static const EdgeKernel kernels[4] = {bwd_src_bf16<false, false>, bwd_src_bf16<true, false>, bwd_src_bf16<false, true>, bwd_src_bf16<true, true>};
  static const EdgeKernel kernels_tr[4] = {bwd_src_bf16_tr<false, false>, bwd_src_bf16_tr<true, false>, bwd_src_bf16_tr<false, true>,
                                           bwd_src_bf16_tr<true, true>};
  // developer switch: AMPCONV_BF16_SRC_TR=0 = the row-softmax kernel (DPP reductions), kept as the cross-check
  static const bool tr = [] {
    const char *e = getenv("AMPCONV_BF16_SRC_TR");
    return !(e && e[0] == '0');
  }();
  return launch(a, L, D, H, tr ? kernels_tr : kernels, stream);
}
