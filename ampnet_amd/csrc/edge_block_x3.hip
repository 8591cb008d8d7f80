// Workgroup-per-unit edge kernels for the shapes of edge_block.hip (L <= 64, even dh <= 64) with fp32 storage (the reference's
// AMPGCN class defaults: L = 40, D = 100, H = 2 -> dh = 50, src/ampnet/module/amp_gcn.py:21-35), OFF the FP32 pipe:
// every fp32 tile is split into THREE bf16 planes on its way into LDS
//      x = h + m + l,   h = bf16(x), m = bf16(x - h), l = bf16(x - h - m)        (24 significand bits, no scale needed:
//                                                                                 bf16 has fp32's exponent range)
// and every product is the fp32 sum of six v_mfma_f32_16x16x32_bf16 partial products
//      a b ~ a_l b_h + a_h b_l + a_m b_m + a_m b_h + a_h b_m + a_h b_h           (dropped: a_m b_l + a_l b_m + a_l b_l
//                                                                                 <= 2^-24 |a b|)
// -- 96 matrix-pipe cycles per 16 x 16 x 32 block instead of the 256 of eight v_mfma_f32_16x16x4_f32, and on the 16-bit
// matrix pipe, which runs beside the vector ALU instead of on it.  Softmax, delta and all sums stay fp32; HBM traffic,
// statistics hand-off and the long-segment plan are those of edge_block.hip (same entry points, same bytes).
//
// One WORKGROUP owns one (row, head) unit; wave w of its NT = ceil(L / 16) owns the unit's own tokens 16 w .. 16 w + 15
// (the COLUMNS of every score tile), all waves share the LDS images of the streamed pair of tiles.  Per plane an image
// is [16 NT token rows][64 channels] bf16 = 128-byte rows (dh <= 32: template KS = 1, one 32-channel k-step and two channel
// tiles instead of two and four; the upper half of every row is then neither staged nor read), 16-byte chunk c of row j stored at chunk c ^ 2 ((j >> 1) & 3):
// the ds_read_b128 of the channel-product fragments and the ds_read_b64_tr_b16 of the token-product fragments are both
// bank-conflict free (tools/lds_bank_check.py).
// Reference arithmetic replaced: torch functional.py:6578-6594 per edge, the mean of amp_conv.py:11, and their autograd
// backward -- as edge_block.hip.
#include "mfma_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2eX = 1.4426950408889634f;
constexpr int kRowB = 128;                 // one plane of a token row: 64 bf16 channels
constexpr int kTileRowsB = 16 * kRowB;     // 16 token rows of one plane

#define MFMA_X3(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)

struct Frag3 {
  i32x4 h, m, l;
};

// the six partial products of one fp32-grade product, smallest first
__device__ __forceinline__ f32x4 mfma6(const Frag3 &a, const Frag3 &b, f32x4 c) {
  c = MFMA_X3(a.l, b.h, c);
  c = MFMA_X3(a.h, b.l, c);
  c = MFMA_X3(a.m, b.m, c);
  c = MFMA_X3(a.m, b.h, c);
  c = MFMA_X3(a.h, b.m, c);
  return MFMA_X3(a.h, b.h, c);
}

__device__ __forceinline__ int pk_bf(float a, float b) {      // v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bfl(int u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bfh(int u) { return __builtin_bit_cast(float, u & (int)0xFFFF0000u); }
// (x0, x1) -> packed bf16 pairs of the three planes (the residuals are exact fp32 differences)
__device__ __forceinline__ void split3(float x0, float x1, int &h, int &m, int &l) {
  h = pk_bf(x0, x1);
  float r0 = x0 - bfl(h), r1 = x1 - bfh(h);
  m = pk_bf(r0, r1);
  r0 -= bfl(m);
  r1 -= bfh(m);
  l = pk_bf(r0, r1);
}

// byte offset of 16-byte chunk `ch` (0..7) of token row j in a plane image
__device__ __forceinline__ int xoff(int j, int ch) { return j * kRowB + ((ch ^ (((j >> 1) & 3) << 1)) << 4); }

struct XArgs {
  ampconv_view_t Q, K, V, dO, O, dK, dV;     // O = forward output / dQ
  const int32_t *ptr, *idx, *qidx;
  const float *cinv;
  const int32_t *spos;
  float *stats;
  HubArgs hub;              // long-segment plan (hub.hip): mode 1 skips long rows, mode 2 = one unit per chunk
  int64_t n_units;
  int L, dh, H;
  float qscale, oscale;
  const float *bounds;      // scaled kernels: device {bound of |Q|K|V|, bound of |dObar|, recorded max |V|, recorded max |dObar|}
  float *absmax;            // scaled backward kernels, or null: atomic max of the finite magnitudes written
};

__device__ __forceinline__ const float *tile_of(const ampconv_view_t &v, int64_t n, int h) {
  return reinterpret_cast<const float *>(v.ptr) + n * v.node_stride + (int64_t)h * v.head_stride;
}

// ---- cooperative staging of two [L x dh] fp32 tiles (A then B): global -> registers -> split -> three plane images
// each.  Thread (r0 = tid / DVP, cv = tid % DVP) owns vector column cv (VEC floats) of rows r0 + i RS.
template <int VEC, int NT, int KS>
struct StageX {
  static constexpr int DVP = 32 * KS / VEC;                 // vector slots per padded row (KS k-steps of 32 channels)
  static constexpr int RS = 64 * NT / DVP;                  // rows per pass
  static constexpr int NP = (16 * NT + RS - 1) / RS;        // passes
  float v[2][NP][VEC];
};

// Loads: raw buffer loads off a per-tile resource (base = the (node, head) tile, uniform; num_records = the bytes of the
// tile that exist), ONE per-thread byte offset per tensor and a scalar offset per pass -- no per-lane address arithmetic,
// no predicates: token rows >= L lie beyond num_records and channels >= dh carry an out-of-range offset, both read as 0
// and are filed as the zero padding of the image.
struct StageSrc {
  unsigned voA, voB;        // byte offset of (row r0, channel c) in a tile of tensor A / B
  int stepA, stepB;         // bytes between two passes (RS token rows)
  int nrecA, nrecB;         // bytes from the tile's first element to the end of its last row
};
template <int VEC, int NT, int KS>
__device__ __forceinline__ StageSrc stage_src(int sA, int sB, int L, int dh, int tid) {
  using S = StageX<VEC, NT, KS>;
  const int cv = tid % S::DVP, r0 = tid / S::DVP, c = cv * VEC;
  StageSrc q;
  q.voA = c < dh ? (unsigned)(r0 * sA + c) * 4u : 0x80000000u;
  q.voB = c < dh ? (unsigned)(r0 * sB + c) * 4u : 0x80000000u;
  q.stepA = S::RS * sA * 4;
  q.stepB = S::RS * sB * 4;
  q.nrecA = ((L - 1) * sA + dh) * 4;
  q.nrecB = ((L - 1) * sB + dh) * 4;
  return q;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const float *base, int nrec) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, nrec, 0x00020000);
}

template <int VEC, int NT, int KS>
__device__ __forceinline__ void xstage_load(StageX<VEC, NT, KS> &s, const float *A, const float *B, const StageSrc &q, int L) {
  using S = StageX<VEC, NT, KS>;
  const __amdgpu_buffer_rsrc_t ra = tile_rsrc(A, q.nrecA), rb = tile_rsrc(B, q.nrecB);
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    if (i * S::RS < L) {                                    // (uniform: whole passes beyond L are not issued)
      if constexpr (VEC == 4) {
        // (bit_cast of the builtin's result as a whole: element access on the returned vector type picks element 0 for
        // every index with this compiler)
        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, q.voA, i * q.stepA, 0));
        const f32x4 y = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, q.voB, i * q.stepB, 0));
        s.v[0][i][0] = x[0]; s.v[0][i][1] = x[1]; s.v[0][i][2] = x[2]; s.v[0][i][3] = x[3];
        s.v[1][i][0] = y[0]; s.v[1][i][1] = y[1]; s.v[1][i][2] = y[2]; s.v[1][i][3] = y[3];
      } else {
        const f32x2 x = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, q.voA, i * q.stepA, 0));
        const f32x2 y = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rb, q.voB, i * q.stepB, 0));
        s.v[0][i][0] = x[0]; s.v[0][i][1] = x[1];
        s.v[1][i][0] = y[0]; s.v[1][i][1] = y[1];
      }
    }
  }
}

// LDS byte offsets of this thread's vector in the rows it stages (plane 0 of an image)
template <int VEC, int NT, int KS>
struct StageOffs {
  int v[StageX<VEC, NT, KS>::NP];
};
template <int VEC, int NT, int KS>
__device__ __forceinline__ void xstage_offsets(StageOffs<VEC, NT, KS> &lo, int tid) {
  using S = StageX<VEC, NT, KS>;
  const int cv = tid % S::DVP, r0 = tid / S::DVP, c = cv * VEC;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) lo.v[i] = xoff(r0 + i * S::RS, c >> 3) + (c & 7) * 2;
}

template <int VEC, int NT, int KS, bool SCALE>
__device__ __forceinline__ void xstage_store(char *imgA, char *imgB, const StageX<VEC, NT, KS> &s,
                                             const StageOffs<VEC, NT, KS> &lo, float mulA, float mulB, int L) {
  using S = StageX<VEC, NT, KS>;
  constexpr int PB = 16 * NT * kRowB;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    if (i * S::RS < L) {
      {                                                     // (lanes beyond the tile hold zeros: the image's padding)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
          char *img = (x ? imgB : imgA) + lo.v[i];
          const float mul = x ? mulB : mulA;
          if constexpr (VEC == 4) {
            int h0, m0, l0, h1, m1, l1;
            if (SCALE) {
              split3(s.v[x][i][0] * mul, s.v[x][i][1] * mul, h0, m0, l0);
              split3(s.v[x][i][2] * mul, s.v[x][i][3] * mul, h1, m1, l1);
            } else {
              split3(s.v[x][i][0], s.v[x][i][1], h0, m0, l0);
              split3(s.v[x][i][2], s.v[x][i][3], h1, m1, l1);
            }
            *reinterpret_cast<i32x2 *>(img) = i32x2{h0, h1};
            *reinterpret_cast<i32x2 *>(img + PB) = i32x2{m0, m1};
            *reinterpret_cast<i32x2 *>(img + 2 * PB) = i32x2{l0, l1};
          } else {
            int h0, m0, l0;
            if (SCALE) split3(s.v[x][i][0] * mul, s.v[x][i][1] * mul, h0, m0, l0);
            else split3(s.v[x][i][0], s.v[x][i][1], h0, m0, l0);
            *reinterpret_cast<int *>(img) = h0;
            *reinterpret_cast<int *>(img + PB) = m0;
            *reinterpret_cast<int *>(img + 2 * PB) = l0;
          }
        }
      }
    }
  }
}

// the unit's own side as COLUMN fragments (B operand), once per unit from global memory: lane (n = lane & 15, kg) holds
// channels 32 ks + 8 kg .. + 7 of token 16 wave + n, scaled, split; token rows >= L and channels >= dh read as zero.
// Two steps, so that the loads are in flight while the unit's first tiles are requested (own_load), and are only waited
// for behind that (own_split).
template <int KS>
struct OwnRaw {
  float2 x[KS][4];
};
template <int KS>
__device__ __forceinline__ void own_load(OwnRaw<KS> &o, const float *base, int row_stride, int wave, int L, int dh, int lane) {
  const int n = lane & 15, kg = lane >> 4, j = 16 * wave + n;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 32 * ks + 8 * kg + 2 * u;
      o.x[ks][u] = make_float2(0.f, 0.f);
      if (j < L && c < dh) o.x[ks][u] = *reinterpret_cast<const float2 *>(base + j * row_stride + c);
    }
  }
}
template <int KS>
__device__ __forceinline__ void own_split(Frag3 (&f)[KS], const OwnRaw<KS> &o, float mul) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    int h[4], m[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) split3(o.x[ks][u].x * mul, o.x[ks][u].y * mul, h[u], m[u], l[u]);
    f[ks].h = i32x4{h[0], h[1], h[2], h[3]};
    f[ks].m = i32x4{m[0], m[1], m[2], m[3]};
    f[ks].l = i32x4{l[0], l[1], l[2], l[3]};
  }
}

// channel-product fragment (A operand) of token tile t, k-step ks: `aks` = this lane's byte offset for that k-step
template <int PB>
__device__ __forceinline__ Frag3 rowfrag3(const char *img, int aks, int t) {
  const char *p = img + aks + t * kTileRowsB;
  return Frag3{*reinterpret_cast<const i32x4 *>(p), *reinterpret_cast<const i32x4 *>(p + PB),
               *reinterpret_cast<const i32x4 *>(p + 2 * PB)};
}

__device__ __forceinline__ i32x2 tr64(const char *p) {
  return __builtin_bit_cast(i32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p));
}
// token-product fragment (A operand, rows = channels 16 mc .. + 15) over the token tiles (2 pair, 2 pair + 1): k-slots
// 0..3 = tokens 16 (2 pair) + 4 kg + 0..3, slots 4..7 = the same rows of the next tile (a last odd tile: its own values
// again -- finite -- against zeros in the other operand).  `trb` = this lane's byte offset for channel tile mc.
template <int NT, int PB>
__device__ __forceinline__ Frag3 colfrag3(const char *img, int trb, int pair) {
  const char *p = img + trb + 2 * pair * kTileRowsB;
  const bool two = 2 * pair + 1 < NT;
  const i32x2 a0 = tr64(p), a1 = tr64(p + PB), a2 = tr64(p + 2 * PB);
  const i32x2 b0 = two ? tr64(p + kTileRowsB) : a0, b1 = two ? tr64(p + kTileRowsB + PB) : a1,
              b2 = two ? tr64(p + kTileRowsB + 2 * PB) : a2;
  return Frag3{i32x4{a0[0], a0[1], b0[0], b0[1]}, i32x4{a1[0], a1[1], b1[0], b1[1]}, i32x4{a2[0], a2[1], b2[0], b2[1]}};
}
// C/D tiles of a tile pair (lane (n, g), reg q of tile t = token 16 t + 4 g + q) -> the B operand of the token product
template <int NT>
__device__ __forceinline__ Frag3 cd_frag3(const f32x4 (&T)[NT], int pair) {
  int h[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
  const f32x4 a = T[2 * pair];
  split3(a[0], a[1], h[0], m[0], l[0]);
  split3(a[2], a[3], h[1], m[1], l[1]);
  if (2 * pair + 1 < NT) {
    const f32x4 b = T[2 * pair + 1 < NT ? 2 * pair + 1 : 0];
    split3(b[0], b[1], h[2], m[2], l[2]);
    split3(b[2], b[3], h[3], m[3], l[3]);
  }
  return Frag3{i32x4{h[0], h[1], h[2], h[3]}, i32x4{m[0], m[1], m[2], m[3]}, i32x4{l[0], l[1], l[2], l[3]}};
}

// every transposed fragment of a product group is in its registers before the group's first MFMA issues, and no
// transposed read is scheduled in among the MFMAs (tests/test_abi.py scans the shipped ISA)
#define X3_FRAG_FENCE()                                    \
  do {                                                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
  } while (0)
// in front of a group of transposed reads that follows MFMAs (their fragment registers may be taken over)
#define X3_PRE_READ()                          \
  do {                                         \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("s_nop 7" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);         \
  } while (0)

// per-lane address parts of the fragment reads
struct FragAddr {
  int a[2];       // channel-product fragments, k-steps 0 / 1
  int tr[4];      // token-product fragments, channel tiles 0..3
};
__device__ __forceinline__ FragAddr frag_addr(int lane) {
  FragAddr fa;
  const int m = lane & 15, kg = lane >> 4;
  fa.a[0] = xoff(m, kg);
  fa.a[1] = fa.a[0] ^ 64;                                   // chunk 4 + kg: the swizzle only touches bits 1..2
  const int q = (lane >> 2) & 3, pp = lane & 3, row = 4 * kg + q;
#pragma unroll
  for (int mc = 0; mc < 4; ++mc) fa.tr[mc] = xoff(row, 2 * mc + (pp >> 1)) + ((pp & 1) << 3);
  return fa;
}

// output: C/D tiles [channel tile mc] of this wave's token tile -> global rows (channels < dh, tokens < L).  Lane
// (token n = lane & 15, g), register r of tile mc = channel 16 mc + 4 g + r
template <int VEC, int MCT>
__device__ __forceinline__ float store_x3(const ampconv_view_t &v, int64_t node, int h, const f32x4 (&T)[MCT], float scale,
                                          int tile, int L, int dh, int lane) {
  const int i = (lane & 15) + 16 * tile, g = lane >> 4;
  float mx = 0.f;                      // largest finite magnitude stored (the scaled backward kernels record it)
  if (i >= L) return mx;
  float *row = reinterpret_cast<float *>(v.ptr) + node * v.node_stride + (int64_t)h * v.head_stride + (int64_t)i * v.row_stride;
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) {
    const int c = 16 * mc + 4 * g;
    const float x0 = T[mc][0] * scale, x1 = T[mc][1] * scale, x2 = T[mc][2] * scale, x3 = T[mc][3] * scale;
    if constexpr (VEC == 4) {
      if (c < dh) {
        *reinterpret_cast<float4 *>(row + c) = make_float4(x0, x1, x2, x3);
        mx = finite_abs_max(mx, make_float4(x0, x1, x2, x3));
      }
    } else {
      if (c < dh) {
        *reinterpret_cast<float2 *>(row + c) = make_float2(x0, x1);
        mx = fmaxf(mx, fmaxf(finite_abs(x0), finite_abs(x1)));
      }
      if (c + 2 < dh) {
        *reinterpret_cast<float2 *>(row + c + 2) = make_float2(x2, x3);
        mx = fmaxf(mx, fmaxf(finite_abs(x2), finite_abs(x3)));
      }
    }
  }
  return mx;
}

// softmax over the source tokens (MFMA rows of every token tile) of one destination-token column; returns m + log2(sum)
template <int NT>
__device__ __forceinline__ float x3_column_softmax(f32x4 (&S)[NT], int L, int g) {
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

// the staging passes rewrite whole 64-channel rows (zeros in the padding) of every token row below RS ceil(L / RS); the
// rows above that, in all plane images, are zeroed once per unit
template <int VEC, int NT, int KS, int NPLANES = 6>
__device__ __forceinline__ void lds_zero_tail(char *p, int L, int tid) {
  using S = StageX<VEC, NT, KS>;
  constexpr int PB = 16 * NT * kRowB;
  const int zr = ((L + S::RS - 1) / S::RS) * S::RS, nrow = 16 * NT - zr;      // rows zr .. 16 NT - 1
  for (int i = tid; i < NPLANES * nrow * (kRowB / 16); i += 64 * NT) {
    const int plane = i / (nrow * (kRowB / 16)), rem = i - plane * (nrow * (kRowB / 16));
    *reinterpret_cast<i32x4 *>(p + plane * PB + zr * kRowB + rem * 16) = i32x4{0, 0, 0, 0};
  }
}

// ---------------------------------------------------------------- forward
#ifndef AMPCONV_X3_FWD_WAVES      // developer A/B switches: minimum waves per SIMD; channel tiles per group of reads
#define AMPCONV_X3_FWD_WAVES 3
#endif
#ifndef AMPCONV_X3_FWD_MCB
#define AMPCONV_X3_FWD_MCB 1
#endif
template <int VEC, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_X3_FWD_WAVES) void fwd_x3(XArgs a) {
  constexpr int MCT = 2 * KS;                 // 16-channel tiles
  constexpr int MCB = AMPCONV_X3_FWD_MCB;
  constexpr int PB = 16 * NT * kRowB, TB = 3 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + TB;
  const int64_t d = a.qidx ? a.qidx[r] : r;

  IdxWindow win;
  if (beg < end) idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);      // (first: everything else waits for it)
  OwnRaw<KS> qraw;
  own_load(qraw, tile_of(a.Q, d, h), (int)a.Q.row_stride, wave, L, dh, lane);
  f32x4 OT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) OT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  const StageSrc sq = stage_src<VEC, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    xstage_load<VEC, NT, KS>(st, tile_of(a.K, s, h), tile_of(a.V, s, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS>(lds, L, tid);
  Frag3 qf[KS];
  own_split(qf, qraw, a.qscale);
  __syncthreads();
  for (int p = beg; p < end; ++p) {
    xstage_store<VEC, NT, KS, false>(Kt, Vt, st, lo, 1.f, 1.f, L);
#ifndef AMPCONV_X3_NOLOADS          // developer probe: the first edge's tiles again and again (what does the compute side cost?)
    if (p + 1 < end) fetch(p + 1);
#endif
    __syncthreads();

#ifdef AMPCONV_X3_NOCOMPUTE          // developer probe: staging, LDS images and barriers only (what does the memory side cost?)
    OT[0][0] += *reinterpret_cast<const float *>(Kt + 4 * tid) + *reinterpret_cast<const float *>(Vt + 4 * tid);
    __syncthreads();
    continue;
#endif
    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) S[t] = mfma6(rowfrag3<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
    }
    x3_column_softmax<NT>(S, L, g);
    Frag3 pf[NPAIR];
#pragma unroll
    for (int mb = 0; mb < MCT / MCB; ++mb) {                  // MCB channel tiles per group of transposed reads
      X3_PRE_READ();
      Frag3 vc[MCB][NPAIR];
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) vc[u][i] = colfrag3<NT, PB>(Vt, fa.tr[MCB * mb + u], i);
      if (mb == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) pf[i] = cd_frag3<NT>(S, i);      // (in the shadow of the reads)
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) OT[MCB * mb + u] = mfma6(vc[u][i], pf[i], OT[MCB * mb + u]);
    }
    __syncthreads();
  }
  // hub pass: unnormalised partial tile, the combine pass applies 1/deg
  store_x3<VEC, MCT>(a.O, onode, h, OT, a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f), wave, L, dh, lane);
}

// ---------------------------------------------------------------- backward, destination pass
#ifndef AMPCONV_X3_DST_WAVES
#define AMPCONV_X3_DST_WAVES 2
#endif
#ifndef AMPCONV_X3_DST_MCB
#define AMPCONV_X3_DST_MCB 2
#endif
#ifndef AMPCONV_X3_SRC_WAVES
#define AMPCONV_X3_SRC_WAVES 2
#endif
#ifndef AMPCONV_X3_SRC_PAIRS
#define AMPCONV_X3_SRC_PAIRS 0
#endif
template <int VEC, bool STATS, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_X3_DST_WAVES) void bwd_dst_x3(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int MCB = AMPCONV_X3_DST_MCB;
  constexpr int PB = 16 * NT * kRowB, TB = 3 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + TB;
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN

  IdxWindow win;
  const float *wts = reinterpret_cast<const float *>(a.spos);
  if (beg < end) idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
  OwnRaw<KS> qraw, graw;
  own_load(qraw, tile_of(a.Q, r, h), (int)a.Q.row_stride, wave, L, dh, lane);
  own_load(graw, tile_of(a.dO, r, h), (int)a.dO.row_stride, wave, L, dh, lane);
  f32x4 dQT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dQT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  float pos_next = 0.f;                      // STATS: CSC position (int bits) in the window's weight slot
  const StageSrc sq = stage_src<VEC, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, wts, p, end, lane, &pos_next);
    xstage_load<VEC, NT, KS>(st, tile_of(a.K, s, h), tile_of(a.V, s, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS>(lds, L, tid);
  Frag3 qf[KS], gf[KS];
  own_split(qf, qraw, a.qscale);
  own_split(gf, graw, inv);
  __syncthreads();
  constexpr int LS = 16 * NT;
  for (int p = beg; p < end; ++p) {
    xstage_store<VEC, NT, KS, false>(Kt, Vt, st, lo, 1.f, 1.f, L);
    float *sb = nullptr;
    if (STATS) sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos_next) * a.H + h) * (2 * LS);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT], dP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        S[t] = mfma6(rowfrag3<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
        dP[t] = mfma6(rowfrag3<PB>(Vt, fa.a[ks], t), gf[ks], dP[t]);
      }
    }
    const float lse = x3_column_softmax<NT>(S, L, g);
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
      for (int q = 0; q < 4; ++q) S[t][q] *= dP[t][q] - delta;       // dS
    }
    Frag3 sf[NPAIR];
#pragma unroll
    for (int mb = 0; mb < MCT / MCB; ++mb) {
      X3_PRE_READ();
      Frag3 kc[MCB][NPAIR];
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) kc[u][i] = colfrag3<NT, PB>(Kt, fa.tr[MCB * mb + u], i);
      if (mb == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) sf[i] = cd_frag3<NT>(S, i);
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) dQT[MCB * mb + u] = mfma6(kc[u][i], sf[i], dQT[MCB * mb + u]);
    }
    __syncthreads();
  }
  store_x3<VEC, MCT>(a.O, onode, h, dQT, a.hub.mode == 2 ? 1.f : a.oscale, wave, L, dh, lane);
}

// ---------------------------------------------------------------- backward, source pass (needs the statistics)
template <int VEC, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_X3_SRC_WAVES) void bwd_src_x3(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int PB = 16 * NT * kRowB, TB = 3 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t s, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4, n = lane & 15;
  char *Qt = lds, *Gt = lds + TB;

  IdxWindow win;
  if (beg < end) idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
  OwnRaw<KS> kraw, vraw;
  own_load(kraw, tile_of(a.K, s, h), (int)a.K.row_stride, wave, L, dh, lane);
  own_load(vraw, tile_of(a.V, s, h), (int)a.V.row_stride, wave, L, dh, lane);
  f32x4 dKT[MCT], dVT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dKT[mc] = dVT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  float inv_next = 0.f;
  // the edge's softmax statistics (2 LS floats, written by the destination pass at this CSC position) travel with its
  // tiles: one float per thread, requested an edge ahead and handed to the waves through LDS
  constexpr int LS = 16 * NT;
  float *sl = reinterpret_cast<float *>(lds + 2 * TB);
  float stat_next = 0.f;
  const StageSrc sq = stage_src<VEC, NT, KS>((int)a.Q.row_stride, (int)a.dO.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv_next);
    if (tid < 2 * LS) stat_next = a.stats[((int64_t)p * a.H + h) * (2 * LS) + tid];
    xstage_load<VEC, NT, KS>(st, tile_of(a.Q, d, h), tile_of(a.dO, d, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS>(lds, L, tid);
  Frag3 kf[KS], vf[KS];
  own_split(kf, kraw, 1.f);
  own_split(vf, vraw, 1.f);
  __syncthreads();
  const bool colok = n + 16 * wave < L;      // this lane's source token exists
  for (int p = beg; p < end; ++p) {
    xstage_store<VEC, NT, KS, true>(Qt, Gt, st, lo, a.qscale, inv_next, L);
    if (tid < 2 * LS) sl[tid] = stat_next;
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

#if AMPCONV_X3_SRC_PAIRS
    // one pair of destination-token tiles at a time (registers; see bwd_src_xh)
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      f32x4 P[2], dS[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * i + u;
        P[u] = dS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < NT) {
          const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sl + 16 * t + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sl + LS + 16 * t + 4 * g);
          f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            S = mfma6(rowfrag3<PB>(Qt, fa.a[ks], t), kf[ks], S);
            dP = mfma6(rowfrag3<PB>(Gt, fa.a[ks], t), vf[ks], dP);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float pr = colok ? fast_exp2(S[q] - l4[q]) : 0.f;
            P[u][q] = pr;
            dS[u][q] = pr * (dP[q] - d4[q]);
          }
        }
      }
      Frag3 pf, sf;
#pragma unroll
      for (int mc = 0; mc < MCT; ++mc) {
        X3_PRE_READ();
        const Frag3 gc = colfrag3<NT, PB>(Gt, fa.tr[mc], i), qc = colfrag3<NT, PB>(Qt, fa.tr[mc], i);
        if (mc == 0) {
          pf = cd_frag3<2>(P, 0);
          sf = cd_frag3<2>(dS, 0);
        }
        X3_FRAG_FENCE();
        dVT[mc] = mfma6(gc, pf, dVT[mc]);
        dKT[mc] = mfma6(qc, sf, dKT[mc]);
      }
    }
#else
    f32x4 P[NT], dS[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {             // destination tokens 16 t + 4 g + q
      const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sl + 16 * t + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sl + LS + 16 * t + 4 * g);
      f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        S = mfma6(rowfrag3<PB>(Qt, fa.a[ks], t), kf[ks], S);
        dP = mfma6(rowfrag3<PB>(Gt, fa.a[ks], t), vf[ks], dP);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float pr = colok ? fast_exp2(S[q] - l4[q]) : 0.f;
        P[t][q] = pr;
        dS[t][q] = pr * (dP[q] - d4[q]);
      }
    }
    Frag3 pf[NPAIR], sf[NPAIR];
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc) {
      X3_PRE_READ();
      Frag3 gc[NPAIR], qc[NPAIR];
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        gc[i] = colfrag3<NT, PB>(Gt, fa.tr[mc], i);
        qc[i] = colfrag3<NT, PB>(Qt, fa.tr[mc], i);
      }
      if (mc == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
          pf[i] = cd_frag3<NT>(P, i);
          sf[i] = cd_frag3<NT>(dS, i);
        }
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        dVT[mc] = mfma6(gc[i], pf[i], dVT[mc]);
        dKT[mc] = mfma6(qc[i], sf[i], dKT[mc]);
      }
    }
#endif
    __syncthreads();
  }
  store_x3<VEC, MCT>(a.dK, onode, h, dKT, a.hub.mode == 2 ? 1.f : a.oscale, wave, L, dh, lane);
  store_x3<VEC, MCT>(a.dV, onode, h, dVT, 1.f, wave, L, dh, lane);
}

// =====================================================================================================================
// The same three passes on TWO fp16 planes of the power-of-two-SCALED value (entry points ampconv_*_edge_scaled): with
// a device-side bound of the operands' magnitudes -- the a-priori bound of the projection that produced them, as for the
// plane-format kernels of edge_mfma_f16x2.hip -- x 2^e = hi + lo with |x 2^e| < 2^15 needs two 16-bit
// planes instead of three and three partial products instead of six:
//      a b ~ a_lo b_hi + a_hi b_lo + a_hi b_hi        (dropped: a_lo b_lo <= 2^-22 |a b|)
// Half the matrix-pipe cycles, two thirds of the split's vector instructions and of the LDS traffic of the bf16 version.
// The scale bookkeeping is that of edge_mfma_f16x2.hip: scores meet their scale inside the exponential, P is split as
// P 2^14, dS with one power of two per WAVE from what the wave can see (Cauchy-Schwarz: its own side's largest token-row
// norm, sqrt(64) x the recorded maximum of the streamed tensor); dObar is divided by the in-degree here (own side: once
// per unit, streamed side: in the staging pass), the statistics hand-off carries delta in the units of dP' = dO' V'^T.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define MFMA_XH(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), (c), 0, 0, 0)

constexpr float kPScaleX = 16384.f, kPUnscaleX = 1.f / 16384.f;
constexpr float kMaskedX = -__builtin_inff();     // (scores meet their scale inside the exponential: a finite mask could be scaled back into range)

struct Frag2 {
  i32x4 h, l;
};
__device__ __forceinline__ f32x4 mfma3h(const Frag2 &a, const Frag2 &b, f32x4 c) {
  c = MFMA_XH(a.l, b.h, c);
  c = MFMA_XH(a.h, b.l, c);
  return MFMA_XH(a.h, b.h, c);
}
// 2^(14 - floor(log2 bound)), exponent field clamped (the function of proj_gemm.hip / edge_mfma_f16x2.hip)
__device__ __forceinline__ float plane_scale_x(float bound) {
  int e = (int)((__builtin_bit_cast(unsigned, bound) >> 23) & 0xFFu);
  e = e < 15 ? 15 : (e > 254 ? 254 : e);
  return __builtin_bit_cast(float, (unsigned)(268 - e) << 23);
}
__device__ __forceinline__ int pk_h(float a, float b) {      // v_cvt_pk_f16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(int, __builtin_convertvector(v, f16x2));
}
// (x0, x1), already scaled, |x| < 2^16 -> packed fp16 pairs of the two planes
__device__ __forceinline__ void split2h(float x0, float x1, int &h, int &l) {
  h = pk_h(x0, x1);
  const f16x2 hv = __builtin_bit_cast(f16x2, h);
  l = pk_h(x0 - (float)hv[0], x1 - (float)hv[1]);
}
__device__ __forceinline__ float frag_sumsq_h(const i32x4 &f) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f16x2 v = __builtin_bit_cast(f16x2, f[k]);
    s = __builtin_amdgcn_fdot2(v, v, s, false);
  }
  return s;
}
// largest token-row norm^2 of the wave's own 16 tokens (hi planes of its two k-step fragments): wave-uniform
template <int KS>
__device__ __forceinline__ float own_max_norm2(const Frag2 (&f)[KS]) {
  float q = frag_sumsq_h(f[0].h);
  if constexpr (KS == 2) q += frag_sumsq_h(f[1].h);
  const float t = groups_sum(q);
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row16_max(t))));
}
// scale of dS for a wave whose own side has largest row norm^2 `own2` and whose streamed side is bounded by `other` per
// element (both in plane units, 64 channels): dS * scale < 2^15
__device__ __forceinline__ float ds_scale_x(float own2, float other) {
  return plane_scale_x(2.f * __builtin_sqrtf(own2) * (8.f * other));
}

template <int VEC, int NT, int KS>
__device__ __forceinline__ void xstage_store_h(char *imgA, char *imgB, const StageX<VEC, NT, KS> &s,
                                               const StageOffs<VEC, NT, KS> &lo, float mulA, float mulB, int L) {
  using S = StageX<VEC, NT, KS>;
  constexpr int PB = 16 * NT * kRowB;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    if (i * S::RS < L) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        char *img = (x ? imgB : imgA) + lo.v[i];
        const float mul = x ? mulB : mulA;
        if constexpr (VEC == 4) {
          int h0, l0, h1, l1;
          split2h(s.v[x][i][0] * mul, s.v[x][i][1] * mul, h0, l0);
          split2h(s.v[x][i][2] * mul, s.v[x][i][3] * mul, h1, l1);
          *reinterpret_cast<i32x2 *>(img) = i32x2{h0, h1};
          *reinterpret_cast<i32x2 *>(img + PB) = i32x2{l0, l1};
        } else {
          int h0, l0;
          split2h(s.v[x][i][0] * mul, s.v[x][i][1] * mul, h0, l0);
          *reinterpret_cast<int *>(img) = h0;
          *reinterpret_cast<int *>(img + PB) = l0;
        }
      }
    }
  }
}
template <int KS>
__device__ __forceinline__ void own_split_h(Frag2 (&f)[KS], const OwnRaw<KS> &o, float mul) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    int h[4], l[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) split2h(o.x[ks][u].x * mul, o.x[ks][u].y * mul, h[u], l[u]);
    f[ks].h = i32x4{h[0], h[1], h[2], h[3]};
    f[ks].l = i32x4{l[0], l[1], l[2], l[3]};
  }
}
template <int PB>
__device__ __forceinline__ Frag2 rowfrag2(const char *img, int aks, int t) {
  const char *p = img + aks + t * kTileRowsB;
  return Frag2{*reinterpret_cast<const i32x4 *>(p), *reinterpret_cast<const i32x4 *>(p + PB)};
}
template <int NT, int PB>
__device__ __forceinline__ Frag2 colfrag2(const char *img, int trb, int pair) {
  const char *p = img + trb + 2 * pair * kTileRowsB;
  const bool two = 2 * pair + 1 < NT;
  const i32x2 a0 = tr64(p), a1 = tr64(p + PB);
  const i32x2 b0 = two ? tr64(p + kTileRowsB) : a0, b1 = two ? tr64(p + kTileRowsB + PB) : a1;
  return Frag2{i32x4{a0[0], a0[1], b0[0], b0[1]}, i32x4{a1[0], a1[1], b1[0], b1[1]}};
}
// C/D tiles of a tile pair (already in the split's units, |x| < 2^16) -> the B operand of the token product
template <int NT>
__device__ __forceinline__ Frag2 cd_frag2h(const f32x4 (&T)[NT], int pair) {
  int h[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
  const f32x4 a = T[2 * pair];
  split2h(a[0], a[1], h[0], l[0]);
  split2h(a[2], a[3], h[1], l[1]);
  if (2 * pair + 1 < NT) {
    const f32x4 b = T[2 * pair + 1 < NT ? 2 * pair + 1 : 0];
    split2h(b[0], b[1], h[2], l[2]);
    split2h(b[2], b[3], h[3], l[3]);
  }
  return Frag2{i32x4{h[0], h[1], h[2], h[3]}, i32x4{l[0], l[1], l[2], l[3]}};
}
// softmax over the source tokens of one destination-token column on RAW scores: `sc` = log2e / sqrt(dh) / (scale of
// Q' K'^T) is applied here, the weights leave multiplied by `mul`.  Returns m sc + log2(sum): P = exp2(S' sc - that).
template <int NT>
__device__ __forceinline__ float xh_column_softmax(f32x4 (&S)[NT], float sc, float mul, int L, int g) {
  float m = kMaskedX;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (16 * t + 4 * g + q >= L) S[t][q] = kMaskedX;
      m = fmaxf(m, S[t][q]);
    }
  }
  m = groups_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      S[t][q] = fast_exp2((S[t][q] - m) * sc);
      l += S[t][q];
    }
  }
  l = groups_sum(l);
  const float inv = fast_rcp(l) * mul;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) S[t][q] *= inv;
  }
  return fmaf(m, sc, __builtin_amdgcn_logf(l));
}

#ifndef AMPCONV_XH_FWD_MCB
#define AMPCONV_XH_FWD_MCB 1
#endif
#ifndef AMPCONV_XH_DST_MCB
#define AMPCONV_XH_DST_MCB 1
#endif
#ifndef AMPCONV_XH_FWD_WAVES
#define AMPCONV_XH_FWD_WAVES 4
#endif
#ifndef AMPCONV_XH_DST_WAVES
#define AMPCONV_XH_DST_WAVES 3
#endif
#ifndef AMPCONV_XH_SRC_WAVES
#define AMPCONV_XH_SRC_WAVES 3
#endif
#ifndef AMPCONV_XH_SRC_PAIRS
#define AMPCONV_XH_SRC_PAIRS 1
#endif

// ---------------------------------------------------------------- forward (scaled)
template <int VEC, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_XH_FWD_WAVES) void fwd_xh(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int MCB = AMPCONV_XH_FWD_MCB;
  constexpr int PB = 16 * NT * kRowB, TB = 2 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + TB;
  const float sq = plane_scale_x(a.bounds[0]), uq = 1.f / sq;      // (powers of two: exact)
  const float sc = (a.qscale * uq) * uq;                            // log2e / sqrt(dh) / (scale of Q' K'^T)

  IdxWindow win;
  if (beg < end) idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
  OwnRaw<KS> qraw;
  own_load(qraw, tile_of(a.Q, r, h), (int)a.Q.row_stride, wave, L, dh, lane);
  f32x4 OT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) OT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  const StageSrc sq_ = stage_src<VEC, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    xstage_load<VEC, NT, KS>(st, tile_of(a.K, s, h), tile_of(a.V, s, h), sq_, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS, 4>(lds, L, tid);
  Frag2 qf[KS];
  own_split_h(qf, qraw, sq);
  __syncthreads();
  for (int p = beg; p < end; ++p) {
    xstage_store_h<VEC, NT, KS>(Kt, Vt, st, lo, sq, sq, L);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) S[t] = mfma3h(rowfrag2<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
    }
    xh_column_softmax<NT>(S, sc, kPScaleX, L, g);
    Frag2 pf[NPAIR];
#pragma unroll
    for (int mb = 0; mb < MCT / MCB; ++mb) {
      X3_PRE_READ();
      Frag2 vc[MCB][NPAIR];
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) vc[u][i] = colfrag2<NT, PB>(Vt, fa.tr[MCB * mb + u], i);
      if (mb == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) pf[i] = cd_frag2h<NT>(S, i);      // (in the shadow of the reads)
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) OT[MCB * mb + u] = mfma3h(vc[u][i], pf[i], OT[MCB * mb + u]);
    }
    __syncthreads();
  }
  store_x3<VEC, MCT>(a.O, onode, h, OT, kPUnscaleX * uq * (a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f)), wave, L, dh,
                lane);
}

// ---------------------------------------------------------------- backward, destination pass (scaled)
template <int VEC, bool STATS, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_XH_DST_WAVES) void bwd_dst_xh(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int MCB = AMPCONV_XH_DST_MCB;
  constexpr int PB = 16 * NT * kRowB, TB = 2 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + TB;
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN
  const float sq = plane_scale_x(a.bounds[0]), uq = 1.f / sq, sg = plane_scale_x(a.bounds[1]), ug = 1.f / sg;
  const float sc = (a.qscale * uq) * uq;

  IdxWindow win;
  const float *wts = reinterpret_cast<const float *>(a.spos);
  if (beg < end) idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
  OwnRaw<KS> qraw, graw;
  own_load(qraw, tile_of(a.Q, r, h), (int)a.Q.row_stride, wave, L, dh, lane);
  own_load(graw, tile_of(a.dO, r, h), (int)a.dO.row_stride, wave, L, dh, lane);
  f32x4 dQT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dQT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  float pos_next = 0.f;                      // STATS: CSC position (int bits) in the window's weight slot
  const StageSrc sq_ = stage_src<VEC, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, wts, p, end, lane, &pos_next);
    xstage_load<VEC, NT, KS>(st, tile_of(a.K, s, h), tile_of(a.V, s, h), sq_, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS, 4>(lds, L, tid);
  Frag2 qf[KS], gf[KS];
  own_split_h(qf, qraw, sq);
  own_split_h(gf, graw, inv * sg);
  // dS = P (dP' - delta'), |dP'_ij| <= |dO'_i| |V'_j|: this wave's dO' rows, any V' row
  const float sd = ds_scale_x(own_max_norm2(gf), a.bounds[2] * sq), usd = 1.f / sd;
  __syncthreads();
  constexpr int LS = 16 * NT;
  for (int p = beg; p < end; ++p) {
    xstage_store_h<VEC, NT, KS>(Kt, Vt, st, lo, sq, sq, L);
    float *sb = nullptr;
    if (STATS) sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos_next) * a.H + h) * (2 * LS);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT], dP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        S[t] = mfma3h(rowfrag2<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
        dP[t] = mfma3h(rowfrag2<PB>(Vt, fa.a[ks], t), gf[ks], dP[t]);
      }
    }
    const float lse2 = xh_column_softmax<NT>(S, sc, 1.f, L, g);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[t][q], dP[t][q], part);
    }
    const float delta = groups_sum(part);
    if (STATS && g == 0) {                   // all LS columns: the source pass reads every one
      sb[(lane & 15) + 16 * wave] = lse2;
      sb[LS + (lane & 15) + 16 * wave] = delta;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) S[t][q] *= (dP[t][q] - delta) * sd;       // dS in the split's units
    }
    Frag2 sf[NPAIR];
#pragma unroll
    for (int mb = 0; mb < MCT / MCB; ++mb) {
      X3_PRE_READ();
      Frag2 kc[MCB][NPAIR];
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) kc[u][i] = colfrag2<NT, PB>(Kt, fa.tr[MCB * mb + u], i);
      if (mb == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) sf[i] = cd_frag2h<NT>(S, i);
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int u = 0; u < MCB; ++u)
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) dQT[MCB * mb + u] = mfma3h(kc[u][i], sf[i], dQT[MCB * mb + u]);
    }
    __syncthreads();
  }
  // partial tiles of a long row leave in the units the combine pass expects (it applies 1 / sqrt(dh))
  const float mx = store_x3<VEC, MCT>(a.O, onode, h, dQT, (a.hub.mode == 2 ? 1.f : a.oscale) * (((uq * uq) * ug) * usd), wave, L, dh,
                                 lane);
  if (a.absmax) wave_record_absmax(a.absmax, mx);
}

// ---------------------------------------------------------------- backward, source pass (scaled; needs the statistics)
template <int VEC, int NT, int KS>
__global__ __launch_bounds__(64 * NT, AMPCONV_XH_SRC_WAVES) void bwd_src_xh(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int PB = 16 * NT * kRowB, TB = 2 * PB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t s, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4, n = lane & 15;
  char *Qt = lds, *Gt = lds + TB;
  const float sq = plane_scale_x(a.bounds[0]), uq = 1.f / sq, sg = plane_scale_x(a.bounds[1]), ug = 1.f / sg;
  const float sc = (a.qscale * uq) * uq;

  IdxWindow win;
  if (beg < end) idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
  OwnRaw<KS> kraw, vraw;
  own_load(kraw, tile_of(a.K, s, h), (int)a.K.row_stride, wave, L, dh, lane);
  own_load(vraw, tile_of(a.V, s, h), (int)a.V.row_stride, wave, L, dh, lane);
  f32x4 dKT[MCT], dVT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dKT[mc] = dVT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  StageOffs<VEC, NT, KS> lo;
  xstage_offsets<VEC, NT, KS>(lo, tid);

  StageX<VEC, NT, KS> st;
  float inv_next = 0.f;
  constexpr int LS = 16 * NT;
  float *sl = reinterpret_cast<float *>(lds + 2 * TB);
  float stat_next = 0.f;
  const StageSrc sq_ = stage_src<VEC, NT, KS>((int)a.Q.row_stride, (int)a.dO.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv_next);
    if (tid < 2 * LS) stat_next = a.stats[((int64_t)p * a.H + h) * (2 * LS) + tid];
    xstage_load<VEC, NT, KS>(st, tile_of(a.Q, d, h), tile_of(a.dO, d, h), sq_, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail<VEC, NT, KS, 4>(lds, L, tid);
  Frag2 kf[KS], vf[KS];
  own_split_h(kf, kraw, sq);
  own_split_h(vf, vraw, sq);
  // |dP'_ij| <= |dO'_i| |V'_j|: any dO' row (its elements are bounded by the recorded maximum), this wave's V' rows
  const float sd = ds_scale_x(own_max_norm2(vf), a.bounds[3] * sg), usd = 1.f / sd;
  __syncthreads();
  const bool colok = n + 16 * wave < L;      // this lane's source token exists
  for (int p = beg; p < end; ++p) {
    xstage_store_h<VEC, NT, KS>(Qt, Gt, st, lo, sq, inv_next * sg, L);
    if (tid < 2 * LS) sl[tid] = stat_next;
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

#if AMPCONV_XH_SRC_PAIRS
    // one pair of destination-token tiles at a time: scores, weights and their split live for one pair only (registers:
    // three waves per SIMD), at the price of a group of transposed reads per (pair, channel tile)
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      f32x4 P[2], dS[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * i + u;
        P[u] = dS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < NT) {
          const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sl + 16 * t + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sl + LS + 16 * t + 4 * g);
          f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            S = mfma3h(rowfrag2<PB>(Qt, fa.a[ks], t), kf[ks], S);
            dP = mfma3h(rowfrag2<PB>(Gt, fa.a[ks], t), vf[ks], dP);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float pr = colok ? fast_exp2(fmaf(S[q], sc, -l4[q])) : 0.f;
            P[u][q] = pr * kPScaleX;
            dS[u][q] = (pr * sd) * (dP[q] - d4[q]);
          }
        }
      }
      Frag2 pf, sf;
#pragma unroll
      for (int mc = 0; mc < MCT; ++mc) {
        X3_PRE_READ();
        const Frag2 gc = colfrag2<NT, PB>(Gt, fa.tr[mc], i), qc = colfrag2<NT, PB>(Qt, fa.tr[mc], i);
        if (mc == 0) {
          pf = cd_frag2h<2>(P, 0);
          sf = cd_frag2h<2>(dS, 0);
        }
        X3_FRAG_FENCE();
        dVT[mc] = mfma3h(gc, pf, dVT[mc]);
        dKT[mc] = mfma3h(qc, sf, dKT[mc]);
      }
    }
#else
    f32x4 P[NT], dS[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {             // destination tokens 16 t + 4 g + q
      const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sl + 16 * t + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sl + LS + 16 * t + 4 * g);
      f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        S = mfma3h(rowfrag2<PB>(Qt, fa.a[ks], t), kf[ks], S);
        dP = mfma3h(rowfrag2<PB>(Gt, fa.a[ks], t), vf[ks], dP);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float pr = colok ? fast_exp2(fmaf(S[q], sc, -l4[q])) : 0.f;
        P[t][q] = pr * kPScaleX;
        dS[t][q] = (pr * sd) * (dP[q] - d4[q]);
      }
    }
    Frag2 pf[NPAIR], sf[NPAIR];
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc) {
      X3_PRE_READ();
      Frag2 gc[NPAIR], qc[NPAIR];
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        gc[i] = colfrag2<NT, PB>(Gt, fa.tr[mc], i);
        qc[i] = colfrag2<NT, PB>(Qt, fa.tr[mc], i);
      }
      if (mc == 0) {
#pragma unroll
        for (int i = 0; i < NPAIR; ++i) {
          pf[i] = cd_frag2h<NT>(P, i);
          sf[i] = cd_frag2h<NT>(dS, i);
        }
      }
      X3_FRAG_FENCE();
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) {
        dVT[mc] = mfma3h(gc[i], pf[i], dVT[mc]);
        dKT[mc] = mfma3h(qc[i], sf[i], dKT[mc]);
      }
    }
#endif
    __syncthreads();
  }
  float mx = store_x3<VEC, MCT>(a.dK, onode, h, dKT, (a.hub.mode == 2 ? 1.f : a.oscale) * (((uq * uq) * ug) * usd), wave, L, dh, lane);
  mx = fmaxf(mx, store_x3<VEC, MCT>(a.dV, onode, h, dVT, kPUnscaleX * ug, wave, L, dh, lane));
  if (a.absmax) wave_record_absmax(a.absmax, mx);
}

// =====================================================================================================================
// bf16 STORAGE (dtype AMPCONV_BF16) on the same structure: the rows are 16-bit already, so a tile is ONE plane copied into
// LDS as it stands and every product is one v_mfma_f32_16x16x32_bf16; softmax weights and dS are rounded to bf16 for their
// second product (the accuracy class of this storage mode: rtol 2e-2; edge_mfma_bf16.hip does the same),
// accumulators, softmax and delta stay fp32.  Nothing is pre-scaled (that would round the operands again): the scores meet
// log2e / sqrt(dh) inside the exponential, 1 / in-degree is applied to dS and to the weights that multiply dObar, and
// delta is handed over in the units of the raw dObar V^T product.  Views: strides in bf16 elements, bases and strides
// even (4-byte pieces) at least.  These kernels replace the fp32-MFMA kernels of edge_block.hip for bf16 rows.
template <int EV, int NT, int KS>
struct StageB {                                             // EV = bf16 elements per lane and load (2 or 4)
  static constexpr int DVP = 32 * KS / EV;
  static constexpr int RS = 64 * NT / DVP;
  static constexpr int NP = (16 * NT + RS - 1) / RS;
  int v[2][NP][EV / 2];
};
template <int EV, int NT, int KS>
__device__ __forceinline__ StageSrc stage_src_b(int sA, int sB, int L, int dh, int tid) {
  using S = StageB<EV, NT, KS>;
  const int cv = tid % S::DVP, r0 = tid / S::DVP, c = cv * EV;
  StageSrc q;
  q.voA = c < dh ? (unsigned)(r0 * sA + c) * 2u : 0x80000000u;
  q.voB = c < dh ? (unsigned)(r0 * sB + c) * 2u : 0x80000000u;
  q.stepA = S::RS * sA * 2;
  q.stepB = S::RS * sB * 2;
  q.nrecA = ((L - 1) * sA + dh) * 2;
  q.nrecB = ((L - 1) * sB + dh) * 2;
  return q;
}
template <int EV, int NT, int KS>
__device__ __forceinline__ void bstage_load(StageB<EV, NT, KS> &s, const void *A, const void *B, const StageSrc &q, int L) {
  using S = StageB<EV, NT, KS>;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(A), 0, q.nrecA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(B), 0, q.nrecB, 0x00020000);
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    if (i * S::RS < L) {
      if constexpr (EV == 4) {
        const i32x2 x = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, q.voA, i * q.stepA, 0));
        const i32x2 y = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rb, q.voB, i * q.stepB, 0));
        s.v[0][i][0] = x[0]; s.v[0][i][1] = x[1];
        s.v[1][i][0] = y[0]; s.v[1][i][1] = y[1];
      } else {
        s.v[0][i][0] = (int)__builtin_amdgcn_raw_buffer_load_b32(ra, q.voA, i * q.stepA, 0);
        s.v[1][i][0] = (int)__builtin_amdgcn_raw_buffer_load_b32(rb, q.voB, i * q.stepB, 0);
      }
    }
  }
}
template <int EV, int NT, int KS>
__device__ __forceinline__ void bstage_offsets(int (&lo)[8], int tid) {
  using S = StageB<EV, NT, KS>;
  static_assert(S::NP <= 8, "passes");
  const int cv = tid % S::DVP, r0 = tid / S::DVP, c = cv * EV;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) lo[i] = xoff(r0 + i * S::RS, c >> 3) + (c & 7) * 2;
}
template <int EV, int NT, int KS>
__device__ __forceinline__ void bstage_store(char *imgA, char *imgB, const StageB<EV, NT, KS> &s, const int (&lo)[8], int L) {
  using S = StageB<EV, NT, KS>;
#pragma unroll
  for (int i = 0; i < S::NP; ++i) {
    if (i * S::RS < L) {
      if constexpr (EV == 4) {
        *reinterpret_cast<i32x2 *>(imgA + lo[i]) = i32x2{s.v[0][i][0], s.v[0][i][1]};
        *reinterpret_cast<i32x2 *>(imgB + lo[i]) = i32x2{s.v[1][i][0], s.v[1][i][1]};
      } else {
        *reinterpret_cast<int *>(imgA + lo[i]) = s.v[0][i][0];
        *reinterpret_cast<int *>(imgB + lo[i]) = s.v[1][i][0];
      }
    }
  }
}
template <int EV, int NT, int KS>
__device__ __forceinline__ void lds_zero_tail_b(char *p, int L, int tid) {
  using S = StageB<EV, NT, KS>;
  constexpr int PB = 16 * NT * kRowB;
  const int zr = ((L + S::RS - 1) / S::RS) * S::RS, nrow = 16 * NT - zr;
  for (int i = tid; i < 2 * nrow * (kRowB / 16); i += 64 * NT) {
    const int plane = i / (nrow * (kRowB / 16)), rem = i - plane * (nrow * (kRowB / 16));
    *reinterpret_cast<i32x4 *>(p + plane * PB + zr * kRowB + rem * 16) = i32x4{0, 0, 0, 0};
  }
}
// the unit's own side: lane (n, kg) = the 8 bf16 channels 32 ks + 8 kg .. + 7 of token 16 wave + n, as they lie in memory
template <int KS>
__device__ __forceinline__ void own_frags_b(i32x4 (&f)[KS], const unsigned short *base, int row_stride, int wave, int L, int dh,
                                            int lane) {
  const int n = lane & 15, kg = lane >> 4, j = 16 * wave + n;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    int w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int c = 32 * ks + 8 * kg + 2 * u;
      w[u] = (j < L && c < dh) ? *reinterpret_cast<const int *>(base + j * row_stride + c) : 0;
    }
    f[ks] = i32x4{w[0], w[1], w[2], w[3]};
  }
}
__device__ __forceinline__ const unsigned short *tile_of_b(const ampconv_view_t &v, int64_t n, int h) {
  return reinterpret_cast<const unsigned short *>(v.ptr) + n * v.node_stride + (int64_t)h * v.head_stride;
}
template <int PB>
__device__ __forceinline__ i32x4 rowfrag1(const char *img, int aks, int t) {
  return *reinterpret_cast<const i32x4 *>(img + aks + t * kTileRowsB);
}
template <int NT>
__device__ __forceinline__ i32x4 colfrag1(const char *img, int trb, int pair) {
  const char *p = img + trb + 2 * pair * kTileRowsB;
  const i32x2 a = tr64(p), b = 2 * pair + 1 < NT ? tr64(p + kTileRowsB) : a;
  return i32x4{a[0], a[1], b[0], b[1]};
}
template <int NT>
__device__ __forceinline__ i32x4 cd_frag1(const f32x4 (&T)[NT], int pair) {
  const f32x4 a = T[2 * pair];
  int h2 = 0, h3 = 0;
  if (2 * pair + 1 < NT) {
    const f32x4 b = T[2 * pair + 1 < NT ? 2 * pair + 1 : 0];
    h2 = pk_bf(b[0], b[1]);
    h3 = pk_bf(b[2], b[3]);
  }
  return i32x4{pk_bf(a[0], a[1]), pk_bf(a[2], a[3]), h2, h3};
}
// output tile -> bf16 rows (main pass) or fp32 partial tiles (long-segment pass)
template <int EV, int MCT>
__device__ __forceinline__ void store_xb(const ampconv_view_t &v, int64_t node, int h, const f32x4 (&T)[MCT], float scale, int tile,
                                         int L, int dh, int lane, bool bf) {
  if (!bf) {
    store_x3<2, MCT>(v, node, h, T, scale, tile, L, dh, lane);
    return;
  }
  const int i = (lane & 15) + 16 * tile, g = lane >> 4;
  if (i >= L) return;
  unsigned short *row = reinterpret_cast<unsigned short *>(v.ptr) + node * v.node_stride + (int64_t)h * v.head_stride +
                        (int64_t)i * v.row_stride;
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) {
    const int c = 16 * mc + 4 * g;
    const int p0 = pk_bf(T[mc][0] * scale, T[mc][1] * scale), p1 = pk_bf(T[mc][2] * scale, T[mc][3] * scale);
    if constexpr (EV == 4) {
      if (c < dh) *reinterpret_cast<i32x2 *>(row + c) = i32x2{p0, p1};
    } else {
      if (c < dh) *reinterpret_cast<int *>(row + c) = p0;
      if (c + 2 < dh) *reinterpret_cast<int *>(row + c + 2) = p1;
    }
  }
}

// ---------------------------------------------------------------- forward (bf16 storage)
template <int EV, int NT, int KS>
__global__ __launch_bounds__(64 * NT, 4) void fwd_xb(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int PB = 16 * NT * kRowB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + PB;
  const int64_t d = a.qidx ? a.qidx[r] : r;

  IdxWindow win;
  if (beg < end) idxwin_load<false>(win, a.idx, nullptr, beg, end, lane);
  i32x4 qf[KS];
  own_frags_b<KS>(qf, tile_of_b(a.Q, d, h), (int)a.Q.row_stride, wave, L, dh, lane);
  f32x4 OT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) OT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  int lo[8];
  bstage_offsets<EV, NT, KS>(lo, tid);
  StageB<EV, NT, KS> st;
  const StageSrc sq = stage_src_b<EV, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<false>(win, a.idx, nullptr, p, end, lane, nullptr);
    bstage_load<EV, NT, KS>(st, tile_of_b(a.K, s, h), tile_of_b(a.V, s, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail_b<EV, NT, KS>(lds, L, tid);
  __syncthreads();
  for (int p = beg; p < end; ++p) {
    bstage_store<EV, NT, KS>(Kt, Vt, st, lo, L);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) S[t] = MFMA_X3(rowfrag1<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
    }
    xh_column_softmax<NT>(S, a.qscale, 1.f, L, g);
    i32x4 pf[NPAIR];
    X3_PRE_READ();
    i32x4 vc[MCT][NPAIR];
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc)
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) vc[mc][i] = colfrag1<NT>(Vt, fa.tr[mc], i);
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) pf[i] = cd_frag1<NT>(S, i);
    X3_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc)
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) OT[mc] = MFMA_X3(vc[mc][i], pf[i], OT[mc]);
    __syncthreads();
  }
  store_xb<EV, MCT>(a.O, onode, h, OT, a.hub.mode == 2 ? 1.f : (deg > 0 ? 1.f / (float)deg : 0.f), wave, L, dh, lane,
                    a.hub.mode != 2);
}

// ---------------------------------------------------------------- backward, destination pass (bf16 storage)
template <int EV, bool STATS, int NT, int KS>
__global__ __launch_bounds__(64 * NT, 3) void bwd_dst_xb(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int PB = 16 * NT * kRowB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t r, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, r, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4;
  char *Kt = lds, *Vt = lds + PB;
  const float inv = deg > 0 ? 1.f / (float)deg : 0.f;       // dO is the gradient of the MEAN

  IdxWindow win;
  const float *wts = reinterpret_cast<const float *>(a.spos);
  if (beg < end) idxwin_load<STATS>(win, a.idx, wts, beg, end, lane);
  i32x4 qf[KS], gf[KS];
  own_frags_b<KS>(qf, tile_of_b(a.Q, r, h), (int)a.Q.row_stride, wave, L, dh, lane);
  own_frags_b<KS>(gf, tile_of_b(a.dO, r, h), (int)a.dO.row_stride, wave, L, dh, lane);
  f32x4 dQT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dQT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  int lo[8];
  bstage_offsets<EV, NT, KS>(lo, tid);
  StageB<EV, NT, KS> st;
  float pos_next = 0.f;
  const StageSrc sq = stage_src_b<EV, NT, KS>((int)a.K.row_stride, (int)a.V.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t s = idxwin_get<STATS>(win, a.idx, wts, p, end, lane, &pos_next);
    bstage_load<EV, NT, KS>(st, tile_of_b(a.K, s, h), tile_of_b(a.V, s, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail_b<EV, NT, KS>(lds, L, tid);
  __syncthreads();
  constexpr int LS = 16 * NT;
  for (int p = beg; p < end; ++p) {
    bstage_store<EV, NT, KS>(Kt, Vt, st, lo, L);
    float *sb = nullptr;
    if (STATS) sb = a.stats + ((int64_t)__builtin_bit_cast(int, pos_next) * a.H + h) * (2 * LS);
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

    f32x4 S[NT], dP[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      S[t] = dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        S[t] = MFMA_X3(rowfrag1<PB>(Kt, fa.a[ks], t), qf[ks], S[t]);
        dP[t] = MFMA_X3(rowfrag1<PB>(Vt, fa.a[ks], t), gf[ks], dP[t]);
      }
    }
    const float lse2 = xh_column_softmax<NT>(S, a.qscale, 1.f, L, g);
    float part = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) part = fmaf(S[t][q], dP[t][q], part);
    }
    const float delta = groups_sum(part);            // in the units of the raw dObar V^T product (no 1 / in-degree)
    if (STATS && g == 0) {
      sb[(lane & 15) + 16 * wave] = lse2;
      sb[LS + (lane & 15) + 16 * wave] = delta;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int q = 0; q < 4; ++q) S[t][q] *= (dP[t][q] - delta) * inv;       // dS
    }
    i32x4 sf[NPAIR];
    X3_PRE_READ();
    i32x4 kc[MCT][NPAIR];
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc)
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) kc[mc][i] = colfrag1<NT>(Kt, fa.tr[mc], i);
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) sf[i] = cd_frag1<NT>(S, i);
    X3_FRAG_FENCE();
#pragma unroll
    for (int mc = 0; mc < MCT; ++mc)
#pragma unroll
      for (int i = 0; i < NPAIR; ++i) dQT[mc] = MFMA_X3(kc[mc][i], sf[i], dQT[mc]);
    __syncthreads();
  }
  store_xb<EV, MCT>(a.O, onode, h, dQT, a.hub.mode == 2 ? 1.f : a.oscale, wave, L, dh, lane, a.hub.mode != 2);
}

// ---------------------------------------------------------------- backward, source pass (bf16 storage; needs the statistics)
template <int EV, int NT, int KS>
__global__ __launch_bounds__(64 * NT, 3) void bwd_src_xb(XArgs a) {
  constexpr int MCT = 2 * KS;
  constexpr int PB = 16 * NT * kRowB, NPAIR = (NT + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int64_t s, onode;
  int h, beg, end, deg;
  const int64_t unit = xcd_unit(blockIdx.x, a.n_units, a.H);
  if (unit < 0 || !map_unit(a.hub, a.ptr, unit, a.n_units, a.H, s, onode, h, beg, end, deg)) return;
  const int L = a.L, dh = a.dh, g = lane >> 4, n = lane & 15;
  char *Qt = lds, *Gt = lds + PB;

  IdxWindow win;
  if (beg < end) idxwin_load<true>(win, a.idx, a.cinv, beg, end, lane);
  i32x4 kf[KS], vf[KS];
  own_frags_b<KS>(kf, tile_of_b(a.K, s, h), (int)a.K.row_stride, wave, L, dh, lane);
  own_frags_b<KS>(vf, tile_of_b(a.V, s, h), (int)a.V.row_stride, wave, L, dh, lane);
  f32x4 dKT[MCT], dVT[MCT];
#pragma unroll
  for (int mc = 0; mc < MCT; ++mc) dKT[mc] = dVT[mc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const FragAddr fa = frag_addr(lane);
  int lo[8];
  bstage_offsets<EV, NT, KS>(lo, tid);
  StageB<EV, NT, KS> st;
  float inv_next = 0.f, inv_cur = 0.f;
  constexpr int LS = 16 * NT;
  float *sl = reinterpret_cast<float *>(lds + 2 * PB);
  float stat_next = 0.f;
  const StageSrc sq = stage_src_b<EV, NT, KS>((int)a.Q.row_stride, (int)a.dO.row_stride, L, dh, tid);
  auto fetch = [&](int p) {
    const int64_t d = idxwin_get<true>(win, a.idx, a.cinv, p, end, lane, &inv_next);
    if (tid < 2 * LS) stat_next = a.stats[((int64_t)p * a.H + h) * (2 * LS) + tid];
    bstage_load<EV, NT, KS>(st, tile_of_b(a.Q, d, h), tile_of_b(a.dO, d, h), sq, L);
  };
  if (beg < end) fetch(beg);
  lds_zero_tail_b<EV, NT, KS>(lds, L, tid);
  __syncthreads();
  const bool colok = n + 16 * wave < L;
  for (int p = beg; p < end; ++p) {
    bstage_store<EV, NT, KS>(Qt, Gt, st, lo, L);
    if (tid < 2 * LS) sl[tid] = stat_next;
    inv_cur = inv_next;                              // 1 / in-degree of THIS edge's destination (fetch overwrites inv_next)
    if (p + 1 < end) fetch(p + 1);
    __syncthreads();

#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      f32x4 P[2], dS[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = 2 * i + u;
        P[u] = dS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t < NT) {
          const f32x4 l4 = *reinterpret_cast<const f32x4 *>(sl + 16 * t + 4 * g);
          const f32x4 d4 = *reinterpret_cast<const f32x4 *>(sl + LS + 16 * t + 4 * g);
          f32x4 S = f32x4{0.f, 0.f, 0.f, 0.f}, dP = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            S = MFMA_X3(rowfrag1<PB>(Qt, fa.a[ks], t), kf[ks], S);
            dP = MFMA_X3(rowfrag1<PB>(Gt, fa.a[ks], t), vf[ks], dP);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float pr = colok ? fast_exp2(fmaf(S[q], a.qscale, -l4[q])) * inv_cur : 0.f;      // P / in-degree
            P[u][q] = pr;
            dS[u][q] = pr * (dP[q] - d4[q]);
          }
        }
      }
      i32x4 pf, sf;
#pragma unroll
      for (int mc = 0; mc < MCT; ++mc) {
        X3_PRE_READ();
        const i32x4 gc = colfrag1<NT>(Gt, fa.tr[mc], i), qc = colfrag1<NT>(Qt, fa.tr[mc], i);
        if (mc == 0) {
          pf = cd_frag1<2>(P, 0);
          sf = cd_frag1<2>(dS, 0);
        }
        X3_FRAG_FENCE();
        dVT[mc] = MFMA_X3(gc, pf, dVT[mc]);
        dKT[mc] = MFMA_X3(qc, sf, dKT[mc]);
      }
    }
    __syncthreads();
  }
  const bool bf = a.hub.mode != 2;
  // (partial tiles of a long column: the combine pass of this family multiplies dK by ln 2 -- its fp32 kernels carry
  // log2e / sqrt(dh) in Q --, so they leave with log2e / sqrt(dh) here)
  store_xb<EV, MCT>(a.dK, onode, h, dKT, a.hub.mode == 2 ? a.oscale * kLog2eX : a.oscale, wave, L, dh, lane, bf);
  store_xb<EV, MCT>(a.dV, onode, h, dVT, 1.f, wave, L, dh, lane, bf);
}

typedef void (*X3Kernel)(XArgs);
template <template <int, int, int> class F, int KS>
X3Kernel x3_pick_ks(int vec, int ntok) {
  switch (ntok) {
    case 1: return vec == 4 ? F<4, 1, KS>::get() : F<2, 1, KS>::get();
    case 2: return vec == 4 ? F<4, 2, KS>::get() : F<2, 2, KS>::get();
    case 3: return vec == 4 ? F<4, 3, KS>::get() : F<2, 3, KS>::get();
    default: return vec == 4 ? F<4, 4, KS>::get() : F<2, 4, KS>::get();
  }
}
// ks = k-steps of 32 channels: 1 for dh <= 32, 2 for dh <= 64
template <template <int, int, int> class F>
X3Kernel x3_pick(int vec, int ntok, int ks) {
  return ks == 1 ? x3_pick_ks<F, 1>(vec, ntok) : x3_pick_ks<F, 2>(vec, ntok);
}
template <int VEC, int NT, int KS> struct XFwd { static X3Kernel get() { return fwd_x3<VEC, NT, KS>; } };
template <int VEC, int NT, int KS> struct XDstS { static X3Kernel get() { return bwd_dst_x3<VEC, true, NT, KS>; } };
template <int VEC, int NT, int KS> struct XDst { static X3Kernel get() { return bwd_dst_x3<VEC, false, NT, KS>; } };
template <int VEC, int NT, int KS> struct XSrc { static X3Kernel get() { return bwd_src_x3<VEC, NT, KS>; } };
template <int VEC, int NT, int KS> struct HFwd { static X3Kernel get() { return fwd_xh<VEC, NT, KS>; } };
template <int VEC, int NT, int KS> struct HDstS { static X3Kernel get() { return bwd_dst_xh<VEC, true, NT, KS>; } };
template <int VEC, int NT, int KS> struct HDst { static X3Kernel get() { return bwd_dst_xh<VEC, false, NT, KS>; } };
template <int VEC, int NT, int KS> struct HSrc { static X3Kernel get() { return bwd_src_xh<VEC, NT, KS>; } };
template <int VEC, int NT, int KS> struct BFwd { static X3Kernel get() { return fwd_xb<VEC, NT, KS>; } };
template <int VEC, int NT, int KS> struct BDstS { static X3Kernel get() { return bwd_dst_xb<VEC, true, NT, KS>; } };
template <int VEC, int NT, int KS> struct BDst { static X3Kernel get() { return bwd_dst_xb<VEC, false, NT, KS>; } };
template <int VEC, int NT, int KS> struct BSrc { static X3Kernel get() { return bwd_src_xb<VEC, NT, KS>; } };

int launch_x3(const XArgs &a, int ntok, X3Kernel k, hipStream_t stream, int extra_bytes = 0, int planes = 3) {
  const int64_t nb = xcd_grid(a.n_units, a.H);
  if (nb > INT32_MAX) return AMPCONV_E_BADARG;
  const dim3 grid((unsigned)nb), block(64 * ntok);
  const size_t shmem = (size_t)2 * planes * 16 * ntok * kRowB + extra_bytes;
  hipLaunchKernelGGL(k, grid, block, shmem, stream, a);
  return ampconv_launch_status();
}

XArgs x3_args(int64_t n_rows, int L, int D, int H) {
  XArgs a{};
  a.L = L; a.dh = D / H; a.H = H;
  a.n_units = n_rows * H;
  a.qscale = kLog2eX / sqrtf((float)a.dh);
  return a;
}

}  // namespace

// developer A/B switch: AMPCONV_BLOCK_X3=0 keeps these shapes on the fp32-MFMA kernels of edge_block.hip
bool ampconv_block_x3_supported(int L, int D, int H, bool bf16) {
  static const bool on = [] {
    const char *e = getenv("AMPCONV_BLOCK_X3");
    return !(e && e[0] == '0');
  }();
  const int dh = D / H;
  (void)bf16;       // fp32 rows: three bf16 planes split in the kernel; bf16 rows: one plane as it lies
  return on && L >= 1 && L <= 64 && dh >= 2 && dh <= 64 && dh % 2 == 0;
}

int ampconv_fwd_edge_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                              const int32_t *col, const int32_t *qidx, int64_t n_rows, int L, int D, int H,
                              ampconv_view_t O, HubArgs hub, int vec, bool bf16, hipStream_t stream) {
  XArgs a = x3_args(n_rows, L, D, H);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.O = O;
  a.ptr = rowptr; a.idx = col; a.qidx = qidx;
  const int ntok = (L + 15) / 16, ks = a.dh > 32 ? 2 : 1;
  if (bf16) return launch_x3(a, ntok, x3_pick<BFwd>(vec, ntok, ks), stream, 0, 1);
  return launch_x3(a, ntok, x3_pick<XFwd>(vec, ntok, ks), stream);
}

int ampconv_bwd_edge_dst_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                                  const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                                  ampconv_view_t dQ, HubArgs hub, StatsArgs sa, int vec, bool bf16, hipStream_t stream) {
  XArgs a = x3_args(n_rows, L, D, H);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.O = dQ;
  a.ptr = rowptr; a.idx = col; a.spos = sa.spos; a.stats = sa.stats;
  a.oscale = 1.f / sqrtf((float)a.dh);
  const int ntok = (L + 15) / 16, ks = a.dh > 32 ? 2 : 1;
  if (bf16) return launch_x3(a, ntok, sa.stats ? x3_pick<BDstS>(vec, ntok, ks) : x3_pick<BDst>(vec, ntok, ks), stream, 0, 1);
  return launch_x3(a, ntok, sa.stats ? x3_pick<XDstS>(vec, ntok, ks) : x3_pick<XDst>(vec, ntok, ks), stream);
}

int ampconv_bwd_edge_src_block_x3(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dO,
                                  const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src, int L,
                                  int D, int H, ampconv_view_t dK, ampconv_view_t dV, HubArgs hub, const float *stats,
                                  int vec, bool bf16, hipStream_t stream) {
  if (!stats) return AMPCONV_E_BADARG;
  XArgs a = x3_args(n_src, L, D, H);
  a.hub = hub;
  a.Q = Q; a.K = K; a.V = V; a.dO = dO; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv; a.stats = const_cast<float *>(stats);
  a.oscale = 0.6931471805599453f;       // dK = ln2 * sum dS^T (Q * log2e / sqrt(dh))
  const int ntok = (L + 15) / 16, ks = a.dh > 32 ? 2 : 1;
  if (bf16) {                           // (Q enters unscaled there: dK = sum dS^T Q / sqrt(dh))
    a.oscale = 1.f / sqrtf((float)a.dh);
    return launch_x3(a, ntok, x3_pick<BSrc>(vec, ntok, ks), stream, 2 * 16 * ntok * (int)sizeof(float), 1);
  }
  return launch_x3(a, ntok, x3_pick<XSrc>(vec, ntok, ks), stream, 2 * 16 * ntok * (int)sizeof(float));
}

// ---------------------------------------------------------------------------------------------------------------------
// C-ABI of the scaled kernels (include/ampconv.h, "edge phase on fp32 views with operand bounds")
namespace {
// views of two-float vectors at least: rows, heads and nodes an even number of elements apart, 8-byte aligned base
int x3_vec(const ampconv_view_t *views, int n, int dh) {
  int vec = dh % 4 == 0 ? 4 : 2;
  for (int i = 0; i < n; ++i) {
    const ampconv_view_t &v = views[i];
    if (!v.ptr) return 0;
    while (vec > 1 && (((uintptr_t)v.ptr % (4 * vec)) || v.node_stride % vec || v.row_stride % vec || v.head_stride % vec))
      vec >>= 1;
  }
  return vec;
}
ampconv_view_t x3_partial_view(void *ws, int64_t tile, int64_t n_chunks, int L, int D, int H) {
  return ampconv_view_t{(float *)ws + tile * n_chunks * L * D, (int64_t)L * D, (int64_t)D, (int64_t)(D / H)};
}
int x3_check(int64_t n, int L, int D, int H, const float *bounds) {
  if (L <= 0 || D <= 0 || H <= 0 || D % H != 0 || n < 0 || !bounds) return AMPCONV_E_BADARG;
  if (!ampconv_scaled_supported(L, D, H)) return AMPCONV_E_DTYPE;
  return AMPCONV_OK;
}
}  // namespace

extern "C" int ampconv_scaled_supported(int L, int D, int H) {
  if (L < 1 || D < 1 || H < 1 || D % H != 0) return 0;
  const int dh = D / H;
  if (L <= 20 && (dh == 32 || dh == 16)) return 0;      // the one-wave-per-unit kernels' shapes (plane format / edge_mfma.hip)
  return L <= 64 && dh >= 2 && dh <= 64 && dh % 2 == 0;
}

extern "C" int ampconv_fwd_edge_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, const int32_t *rowptr,
                                       const int32_t *col, int64_t n_rows, int L, int D, int H, ampconv_view_t O,
                                       const void *hub_plan, int64_t hub_chunks, void *hub_ws, const float *bounds,
                                       void *stream) {
  if (int rc = x3_check(n_rows, L, D, H, bounds)) return rc;
  if (n_rows == 0) return AMPCONV_OK;
  const ampconv_view_t views[] = {Q, K, V, O};
  const int vec = x3_vec(views, 4, D / H), ntok = (L + 15) / 16, ks = D / H > 32 ? 2 : 1;
  if (vec < 2 || !rowptr) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  XArgs a = x3_args(n_rows, L, D, H);
  a.Q = Q; a.K = K; a.V = V; a.O = O;
  a.ptr = rowptr; a.idx = col; a.bounds = bounds;
  if (hub_plan && hub_chunks > 0 && hub_ws) {          // long segments: main + hub + combine
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    if (int rc = launch_x3(a, ntok, x3_pick<HFwd>(vec, ntok, ks), st, 0, 2)) return rc;
    const ampconv_view_t P = x3_partial_view(hub_ws, 0, hub_chunks, L, D, H);
    a.O = P;
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    const ampconv_view_t pviews[] = {Q, K, V, P};
    const int pvec = x3_vec(pviews, 4, D / H);
    if (int rc = launch_x3(a, ntok, x3_pick<HFwd>(pvec, ntok, ks), st, 0, 2)) return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, O, rowptr, L, D, H, 1.f, 0, st);
  }
  return launch_x3(a, ntok, x3_pick<HFwd>(vec, ntok, ks), st, 0, 2);
}

extern "C" int ampconv_bwd_edge_dst_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar,
                                           const int32_t *rowptr, const int32_t *col, int64_t n_rows, int L, int D, int H,
                                           ampconv_view_t dQ, const void *hub_plan, int64_t hub_chunks, void *hub_ws,
                                           const float *bounds, const int32_t *spos, float *stats, float *out_absmax,
                                           void *stream) {
  if (int rc = x3_check(n_rows, L, D, H, bounds)) return rc;
  if (stats && (!spos || (uintptr_t)stats % 16 != 0)) return AMPCONV_E_BADARG;
  if (n_rows == 0) return AMPCONV_OK;
  const ampconv_view_t views[] = {Q, K, V, dObar, dQ};
  const int vec = x3_vec(views, 5, D / H), ntok = (L + 15) / 16, ks = D / H > 32 ? 2 : 1;
  if (vec < 2 || !rowptr) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  XArgs a = x3_args(n_rows, L, D, H);
  a.Q = Q; a.K = K; a.V = V; a.dO = dObar; a.O = dQ;
  a.ptr = rowptr; a.idx = col; a.bounds = bounds; a.absmax = out_absmax;
  a.spos = spos; a.stats = stats;
  a.oscale = 1.f / sqrtf((float)a.dh);
  auto pick = [&](int v) { return stats ? x3_pick<HDstS>(v, ntok, ks) : x3_pick<HDst>(v, ntok, ks); };
  if (hub_plan && hub_chunks > 0 && hub_ws) {
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    if (int rc = launch_x3(a, ntok, pick(vec), st, 0, 2)) return rc;
    const ampconv_view_t P = x3_partial_view(hub_ws, 0, hub_chunks, L, D, H);
    a.O = P;
    a.absmax = nullptr;                                // partial tiles: the combine pass records what it writes
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    const ampconv_view_t pviews[] = {Q, K, V, dObar, P};
    if (int rc = launch_x3(a, ntok, pick(x3_vec(pviews, 5, D / H)), st, 0, 2)) return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)P.ptr, dQ, nullptr, L, D, H, a.oscale, 0, st, out_absmax);
  }
  return launch_x3(a, ntok, pick(vec), st, 0, 2);
}

extern "C" int ampconv_bwd_edge_src_scaled(ampconv_view_t Q, ampconv_view_t K, ampconv_view_t V, ampconv_view_t dObar,
                                           const int32_t *cscptr, const int32_t *crow, const float *cinv, int64_t n_src,
                                           int L, int D, int H, ampconv_view_t dK, ampconv_view_t dV, const void *hub_plan,
                                           int64_t hub_chunks, void *hub_ws, const float *bounds, const float *stats,
                                           float *out_absmax, void *stream) {
  if (int rc = x3_check(n_src, L, D, H, bounds)) return rc;
  if (!stats || (uintptr_t)stats % 16 != 0 || !cinv) return AMPCONV_E_BADARG;      // this pass exists only with the hand-off
  if (n_src == 0) return AMPCONV_OK;
  const ampconv_view_t views[] = {Q, K, V, dObar, dK, dV};
  const int vec = x3_vec(views, 6, D / H), ntok = (L + 15) / 16, ks = D / H > 32 ? 2 : 1;
  if (vec < 2 || !cscptr) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  XArgs a = x3_args(n_src, L, D, H);
  a.Q = Q; a.K = K; a.V = V; a.dO = dObar; a.dK = dK; a.dV = dV;
  a.ptr = cscptr; a.idx = crow; a.cinv = cinv; a.bounds = bounds; a.absmax = out_absmax;
  a.stats = const_cast<float *>(stats);
  a.oscale = 1.f / sqrtf((float)a.dh);
  const int extra = 2 * 16 * ntok * (int)sizeof(float);
  if (hub_plan && hub_chunks > 0 && hub_ws) {
    a.hub = HubArgs{(const int32_t *)hub_plan, 1};
    if (int rc = launch_x3(a, ntok, x3_pick<HSrc>(vec, ntok, ks), st, extra, 2)) return rc;
    const ampconv_view_t PK = x3_partial_view(hub_ws, 0, hub_chunks, L, D, H), PV = x3_partial_view(hub_ws, 1, hub_chunks, L, D, H);
    a.dK = PK;
    a.dV = PV;
    a.absmax = nullptr;
    a.hub.mode = 2;
    a.n_units = hub_chunks * H;
    const ampconv_view_t pviews[] = {Q, K, V, dObar, PK, PV};
    if (int rc = launch_x3(a, ntok, x3_pick<HSrc>(x3_vec(pviews, 6, D / H), ntok, ks), st, extra, 2)) return rc;
    if (int rc = ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PK.ptr, dK, nullptr, L, D, H, a.oscale, 0, st, out_absmax))
      return rc;
    return ampconv_hub_combine(hub_plan, hub_chunks, (const float *)PV.ptr, dV, nullptr, L, D, H, 1.f, 0, st, out_absmax);
  }
  return launch_x3(a, ntok, x3_pick<HSrc>(vec, ntok, ks), st, extra, 2);
}
