// Node phase of AMPConv on the bf16 matrix cores at fp32 accuracy (gfx950).
//
// What it replaces: the packed in-projection and the out-projection of nn.MultiheadAttention
// (torch functional.py:5785-5862 `_in_projection_packed`, :6600 `linear(attn_output, out_proj...)`;
// the reference's own copy: src/ampnet/conv/custom_multihead_attn_forward.py:4070-4077) and their
// autograd backward -- dense [N*L, D] x [D, {D, 3D}] products, once per NODE here.
//
// gfx950 runs fp32-input MFMA at the fp32 VECTOR rate (1/16 of the bf16 MFMA rate, and on the same
// pipe as the VALU: DESIGN.md 4), so an fp32 GEMM library is already at the end of that road
// (rocBLAS: 89 % of 157 TF).  Here every fp32 operand element is written EXACTLY as
//   x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)   (3 x 8 = 24 bits)
// and a product is the fp32 sum of the six bf16 x bf16 partial products of order >= 2^-16,
//   a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// on v_mfma_f32_32x32x16_bf16 (the three dropped terms are <= 3 * 2^-26 |a b|, below one fp32 rounding
// of the product itself; bf16 x bf16 products are exact in the fp32 accumulator).  Six MFMAs of 16x the
// fp32 rate = 2.7x the fp32 peak, and bf16 MFMAs co-execute with the VALU work of the split.
//
// Two kernels:
//   proj_rows  out[M, N] = A[M, K] W^T (+ bias) (* row mask)      A fp32 rows, split on the way into LDS;
//              W pre-split ONCE into an image of ready MFMA fragments (ampconv_proj_weight_image) that is
//              copied to LDS by LDS-DMA (no registers, no address arithmetic, conflict-free by construction)
//   proj_wgrad dW[Na, Nb] = A[M, Na]^T B[M, Nb], colsum(A)         (weight / bias gradients: reduction over
//              the N*L node-token rows; both operands split on the fly, fragments by the hardware-transposing
//              ds_read_b64_tr_b16; deterministic two-step reduction over row slices)
#include <type_traits>
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;

constexpr int kFrag = 1024;          // one MFMA operand fragment: 64 lanes x 8 bf16
constexpr int kTile3 = 3 * kFrag;    // the three planes of one (32-row tile, 16-deep k step)

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32 (RNE)
  f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float lo_as_f32(unsigned h) { return __builtin_bit_cast(float, h << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned h) { return __builtin_bit_cast(float, h & 0xFFFF0000u); }

// (x0, x1) -> packed bf16 pairs of the three planes; x == h1 + h2 + h3 exactly
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

#define MFMA32(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), (c), 0, 0, 0)

// six partial products, smallest first
__device__ __forceinline__ f32x16 mfma6(const i32x4 (&a)[3], const i32x4 (&b)[3], f32x16 c) {
  c = MFMA32(a[2], b[0], c);
  c = MFMA32(a[1], b[1], c);
  c = MFMA32(a[0], b[2], c);
  c = MFMA32(a[1], b[0], c);
  c = MFMA32(a[0], b[1], c);
  c = MFMA32(a[0], b[0], c);
  return c;
}

// LDS-DMA of 16 bytes per lane: LDS destination = wave-uniform `lds_dst` + 16 * lane, source per lane.
// Inline assembly on purpose: behind the builtin hipcc orders every later LDS read after the DMA with
// `s_waitcnt vmcnt(0)` (the whole memory latency); the kernels below wait for their DMAs themselves.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

// ---------------------------------------------------------------------------------------------------
// weight image: for k stage kt (32 deep), 32-column tile n32 of the OUTPUT, k step ks (16 deep), plane p:
//   fragment ((kt * N/32 + n32) * 2 + ks) * 3 + p,  1 KiB, lane l = (r = l & 31, h = l >> 5) holds plane p of
//   B[n32 * 32 + r][kt * 32 + ks * 16 + 8 h + 0..7]      with B[n][k] = W[n * stride_n + k * stride_k]
// i.e. exactly the B operand of v_mfma_f32_32x32x16_bf16, in lane order.
__global__ void weight_image_kernel(const float *__restrict__ W, int64_t sn, int64_t sk, int N, int K,
                                    char *__restrict__ img) {
  const int k8s = K / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * k8s) return;
  const int n = idx / k8s, k8 = idx - n * k8s;
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = W[(int64_t)n * sn + (int64_t)(8 * k8 + j) * sk];
  i32x4 pl[3];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const Pair3 q = split_pair(x[2 * t], x[2 * t + 1]);
    pl[0][t] = q.h1; pl[1][t] = q.h2; pl[2][t] = q.h3;
  }
  const int kt = k8 >> 2, ks = (k8 >> 1) & 1, h = k8 & 1, n32 = n >> 5, r = n & 31;
  char *dst = img + ((size_t)((kt * (N / 32) + n32) * 2 + ks) * 3) * kFrag + (32 * h + r) * 16;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(dst + p * kFrag) = pl[p];
}

// ---------------------------------------------------------------------------------------------------
// proj_rows: 128 x BN output tile per 256-thread workgroup (4 waves as 2 x 2, each 64 x BN/2), K in stages
// of 32.  LDS per stage: A planes 24 KiB (one buffer: the next stage's rows wait in registers, split and
// written between the two barriers), weight fragments BN/32 * 6 KiB (two buffers, filled by LDS-DMA a whole
// compute phase ahead).  2 workgroups per CU cover each other's barriers, split phases and store tails.
struct RowsArgs {
  const float *A;
  int64_t lda;
  int64_t M;
  int K, N;
  const char *wimg;
  const float *bias;          // [N] or null
  const int32_t *rowptr;      // null: no mask; else rows of nodes with an empty CSR segment come out 0
  int L;
  float *out;
  int64_t ldc;
  int row_tiles;              // ceil(M / 128)
};

constexpr int kBM = 128;
constexpr int kStageA = (kBM / 32) * 2 * kTile3;     // 24 KiB

template <int BN>
__global__ __launch_bounds__(256, 2) void proj_rows_kernel(RowsArgs a) {
  constexpr int NTB = BN / 32;                 // 32-column tiles per workgroup
  constexpr int NTW = NTB / 2;                 // ... per wave
  constexpr int kStageB = NTB * 2 * kTile3;
  constexpr int kPieces = kStageB / kFrag;     // 1-KiB DMA pieces per stage
  // one LDS object: [B buffer 0][B buffer 1][A planes][row flags]; the DMA targets stay below 64 KiB
  __shared__ __attribute__((aligned(16))) char smem[2 * kStageB + kStageA + kBM * 4];
  char *sB = smem, *sA = smem + 2 * kStageB;
  float *flags = reinterpret_cast<float *>(smem + 2 * kStageB + kStageA);

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w >> 1, wn = w & 1;
  // XCD-aware order: workgroups b, b + 8, b + 16 ... share an XCD (round-robin dispatch); give them the
  // column tiles of ONE row tile so that its A rows are fetched into that XCD's L2 once
  constexpr int kXcd = 8;
  const int nct = a.N / BN;
  const int b = blockIdx.x, xcd = b % kXcd, i_x = b / kXcd;
  const int rt = (i_x / nct) * kXcd + xcd, ct = i_x % nct;
  if (rt >= a.row_tiles) return;
  const int64_t row0 = (int64_t)rt * kBM;
  const int col0 = ct * BN;
  const int KT = a.K / 32;

  if (t < kBM) {
    float f = 1.f;
    if (a.rowptr) {
      const int64_t m = row0 + t < a.M ? row0 + t : a.M - 1;
      const int64_t node = m / a.L;
      f = a.rowptr[node + 1] != a.rowptr[node] ? 1.f : 0.f;
    }
    flags[t] = f;
  }

  // A rows of this thread: float4 column c of the 32-deep stage, row r0 of each of the four 32-row tiles
  const int c = t & 7, r0 = t >> 3;
  const float *arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t m = row0 + 32 * i + r0;
    m = m < a.M ? m : a.M - 1;
    arow[i] = a.A + m * a.lda + 4 * c;
  }
  // where its split chunks go: fragment (tile i, k step c >> 2), slot 32 h + (r ^ (4 ks + 2 h)), half q
  const int wks = c >> 2, wh = (c >> 1) & 1, wq = c & 1;
  char *wdst = sA + wks * kTile3 + ((32 * wh + (r0 ^ (4 * wks + 2 * wh))) << 4) + 8 * wq;
  // fragment reads: lane (r = lane & 31, h = lane >> 5)
  const int fr = lane & 31, fh = lane >> 5;
  const char *ardp[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
    ardp[ks] = sA + (2 * wm) * 2 * kTile3 + ks * kTile3 + ((32 * fh + (fr ^ (4 * ks + 2 * fh))) << 4);
  const char *brdp = sB + (NTW * wn) * 2 * kTile3 + lane * 16;

  // weight image of this column block: stage kt = kPieces KiB at ((kt * N/32 + col0/32) * 6) KiB
  const char *wsrc = a.wimg + (size_t)(col0 / 32) * 2 * kTile3 + lane * 16;
  const size_t wstage = (size_t)(a.N / 32) * 2 * kTile3;
  const unsigned sB_lds = (unsigned)(uintptr_t)(lds_char *)sB;

  auto dma_stage = [&](int kt) {
    const unsigned dst = sB_lds + (kt & 1) * kStageB;
#pragma unroll
    for (int j = 0; j < kPieces / 4; ++j) {
      const int piece = w + 4 * j;
      dma16(wsrc + (size_t)kt * wstage + piece * kFrag, dst + piece * kFrag);
    }
  };

  f32x16 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 x[4];
  dma_stage(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const float4 *>(arow[i]);

  for (int kt = 0; kt < KT; ++kt) {
    // split this stage's rows and file them as MFMA fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const Pair3 p0 = split_pair(x[i].x, x[i].y), p1 = split_pair(x[i].z, x[i].w);
      char *d = wdst + i * 2 * kTile3;
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kFrag) = i32x2{p0.h2, p1.h2};
      *reinterpret_cast<i32x2 *>(d + 2 * kFrag) = i32x2{p0.h3, p1.h3};
    }
    // this stage's weight fragments were requested one compute phase ago
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + 1 < KT) {
      dma_stage(kt + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const float4 *>(arow[i] + 32 * (kt + 1));
    }
    const char *bcur = brdp + (kt & 1) * kStageB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      i32x4 af[2][3], bf[NTW][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[i][p] = *reinterpret_cast<const i32x4 *>(ardp[ks] + i * 2 * kTile3 + p * kFrag);
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          bf[j][p] = *reinterpret_cast<const i32x4 *>(bcur + (2 * j + ks) * kTile3 + p * kFrag);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = mfma6(af[i], bf[j], acc[i][j]);
    }
    // every wave is done with the A planes (and with weight buffer kt & 1) before they are overwritten
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // epilogue: C/D register e of lane (col = lane & 31, hi = lane >> 5) is row (e & 3) + 8 (e >> 2) + 4 hi
  auto store_tile = [&](auto ragged) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rl0 = (2 * wm + i) * 32 + 4 * fh;
      float fl[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 f4 = *reinterpret_cast<const float4 *>(flags + rl0 + 8 * g);
        fl[4 * g] = f4.x; fl[4 * g + 1] = f4.y; fl[4 * g + 2] = f4.z; fl[4 * g + 3] = f4.w;
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int col = col0 + (NTW * wn + j) * 32 + fr;
        const float bj = a.bias ? a.bias[col] : 0.f;
        float *o = a.out + (row0 + rl0) * a.ldc + col;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int dr = (e & 3) + 8 * (e >> 2);
          const float v = (acc[i][j][e] + bj) * fl[e];
          if (!decltype(ragged)::value || row0 + rl0 + dr < a.M) o[(int64_t)dr * a.ldc] = v;
        }
      }
    }
  };
  if (row0 + kBM <= a.M)            // workgroup-uniform: only the last row tile is ragged
    store_tile(std::false_type{});
  else
    store_tile(std::true_type{});
}


// ---------------------------------------------------------------------------------------------------
// proj_wgrad: dW[Na, Nb] = sum over the rows m of A[m, :]^T B[m, :] (+ column sums of A).  The contraction
// index is the ROW of both inputs, so an MFMA operand fragment (8 consecutive k per lane) is a COLUMN piece
// of the row-major tiles: the planes are stored as they come, [16 rows][TI or TJ columns] bf16, and read
// with ds_read_b64_tr_b16 (4 rows x 16 columns per 16 lanes, delivered column-major).  A 16-byte... 64-byte
// chunk index XORed with (row & 3) makes both the 8-byte plane stores and the transposed reads conflict-free.
// One 128 x TJ tile of dW per 256-thread workgroup and row slice; 16 rows per stage, two LDS buffers, one
// barrier per stage, the next stage's rows in flight in registers.  Partial tiles per slice go to the
// workspace and are added in slice order by wgrad_reduce_kernel (bitwise reproducible, no atomics).
struct WgradArgs {
  const float *A;
  int64_t lda;
  const float *B;
  int64_t ldb;
  int64_t M;
  int Na, Nb;
  const int32_t *rowptr;
  int L;
  float *part;                // [S][Na * Nb + Na]
  int S;
  int64_t rows_per_slice;     // multiple of 16
};

constexpr int kRS = 16;       // rows per stage

__device__ __forceinline__ i32x4 tr_frag(const char *p0, const char *p1) {
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
  const i32x2 ai = __builtin_bit_cast(i32x2, a), bi = __builtin_bit_cast(i32x2, b);
  return i32x4{ai[0], ai[1], bi[0], bi[1]};
}

template <int TJ, bool MASK>
__global__ __launch_bounds__(256, 2) void proj_wgrad_kernel(WgradArgs a) {
  constexpr int TI = 128;
  constexpr int kRowA = TI * 2, kRowB = TJ * 2;             // bytes per image row
  constexpr int kPlaneA = kRS * kRowA, kPlaneB = kRS * kRowB;
  constexpr int kStage = 3 * (kPlaneA + kPlaneB);
  constexpr int NJW = TJ / 64;                              // 32-column tiles of B per wave (2 x 2 waves)
  constexpr int NLB = TJ / 64;                              // float4 loads of B per thread and stage
  __shared__ __attribute__((aligned(16))) char smem[2 * kStage];

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wi = w >> 1, wj = w & 1;
  const int ntj = a.Nb / TJ, ntiles = (a.Na / TI) * ntj;
  constexpr int kXcd = 8;
  const int b = blockIdx.x, xcd = b % kXcd, i_x = b / kXcd;
  const int slice = (i_x / ntiles) * kXcd + xcd, tile = i_x % ntiles;
  if (slice >= a.S) return;
  const int ti = tile / ntj, tj = tile % ntj;
  const int64_t m0 = (int64_t)slice * a.rows_per_slice;
  const int64_t m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
  const int ns = (int)((m1 - m0 + kRS - 1) / kRS);

  // loads: A rows ra, ra + 8 (float4 column ca of 32), B rows rb + 4 i (float4 column cb of TJ / 4)
  const int ca = t & 31, ra = t >> 5;
  constexpr int kColsB4 = TJ / 4;
  const int cb = t % kColsB4, rb = t / kColsB4;
  constexpr int kRowsB = 256 / kColsB4;                     // rows of B covered by one load of the workgroup
  const float *pa = a.A + (int64_t)ti * TI + 4 * ca;
  const float *pb = a.B + (int64_t)tj * TJ + 4 * cb;
  // plane stores: 8 bytes at row r, byte (8 c) ^ ((r & 3) << 6)
  int wa[2], wb[NLB];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = ra + 8 * i;
    wa[i] = r * kRowA + ((8 * ca) ^ ((r & 3) << 6));
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int r = rb + kRowsB * i;
    wb[i] = 3 * kPlaneA + r * kRowB + ((8 * cb) ^ ((r & 3) << 6));
  }
  // transposed reads: lane = (h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3) addresses
  // row 8 h + 4 u + q, columns 32 tile + 16 gi + 4 p .. + 3 (u = 0, 1: the two halves of the 8-deep k group)
  const int fh = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3, fr = lane & 31;
  const int rdA = (8 * fh + q) * kRowA + (q << 6) + 32 * gi + 8 * pp;       // ^ (tile << 6), + 4 * kRowA for u = 1
  const int rdB = 3 * kPlaneA + (8 * fh + q) * kRowB + (q << 6) + 32 * gi + 8 * pp;

  f32x16 acc[2][NJW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);

  float4 xa[2], xb[NLB];
  auto load_stage = [&](int s) {
    const int64_t mb = m0 + (int64_t)s * kRS;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int64_t m = mb + ra + 8 * i;
      const int64_t mc = m < m1 ? m : m1 - 1;
      xa[i] = *reinterpret_cast<const float4 *>(pa + mc * a.lda);
      float f = m < m1 ? 1.f : 0.f;
      if (MASK) {
        const int64_t node = mc / a.L;
        f = a.rowptr[node + 1] != a.rowptr[node] ? f : 0.f;
      }
      if (MASK || s == ns - 1) { xa[i].x *= f; xa[i].y *= f; xa[i].z *= f; xa[i].w *= f; }
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int64_t m = mb + rb + kRowsB * i;
      const int64_t mc = m < m1 ? m : m1 - 1;
      xb[i] = *reinterpret_cast<const float4 *>(pb + mc * a.ldb);
    }
  };
  load_stage(0);

  for (int s = 0; s < ns; ++s) {
    char *buf = smem + (s & 1) * kStage;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      cs.x += xa[i].x; cs.y += xa[i].y; cs.z += xa[i].z; cs.w += xa[i].w;
      const Pair3 p0 = split_pair(xa[i].x, xa[i].y), p1 = split_pair(xa[i].z, xa[i].w);
      char *d = buf + wa[i];
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kPlaneA) = i32x2{p0.h2, p1.h2};
      *reinterpret_cast<i32x2 *>(d + 2 * kPlaneA) = i32x2{p0.h3, p1.h3};
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const Pair3 p0 = split_pair(xb[i].x, xb[i].y), p1 = split_pair(xb[i].z, xb[i].w);
      char *d = buf + wb[i];
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kPlaneB) = i32x2{p0.h2, p1.h2};
      *reinterpret_cast<i32x2 *>(d + 2 * kPlaneB) = i32x2{p0.h3, p1.h3};
    }
    if (s + 1 < ns) load_stage(s + 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    i32x4 af[2][3], bf[NJW][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const char *r0 = buf + (rdA ^ ((2 * wi + i) << 6));
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = tr_frag(r0 + p * kPlaneA, r0 + p * kPlaneA + 4 * kRowA);
    }
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int jt = NJW * wj + j;
      const char *r0 = buf + ((rdB + ((jt >> 2) << 8)) ^ ((jt & 3) << 6));
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[j][p] = tr_frag(r0 + p * kPlaneB, r0 + p * kPlaneB + 4 * kRowB);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJW; ++j) acc[i][j] = mfma6(af[i], bf[j], acc[i][j]);
  }

  // partial tile of this slice
  float *part = a.part + (size_t)slice * ((size_t)a.Na * a.Nb + a.Na);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      float *o = part + (size_t)(ti * TI + (2 * wi + i) * 32 + 4 * fh) * a.Nb + tj * TJ + (NJW * wj + j) * 32 + fr;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(size_t)((e & 3) + 8 * (e >> 2)) * a.Nb] = acc[i][j][e];
    }
  if (tj == 0) {
    // column sums of the A tile: 8 row-threads per float4 column, added through LDS in a fixed order
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float4 *red = reinterpret_cast<float4 *>(smem);
    red[ra * 32 + ca] = cs;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t < 32) {
      float4 sum = red[t];
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        const float4 v = red[r * 32 + t];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      *reinterpret_cast<float4 *>(part + (size_t)a.Na * a.Nb + ti * TI + 4 * t) = sum;
    }
  }
}

// out[e] = sum over the slices, in slice order, of part[s][e]; e < n_dw goes to dW, the rest to colsum
__global__ void wgrad_reduce_kernel(const float *__restrict__ part, int S, int64_t n_dw, int64_t n_all,
                                    float *__restrict__ dW, float *__restrict__ colsum) {
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (e >= n_all) return;
  float4 acc = *reinterpret_cast<const float4 *>(part + e);
  for (int s = 1; s < S; ++s) {
    const float4 v = *reinterpret_cast<const float4 *>(part + (size_t)s * n_all + e);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (e < n_dw)
    *reinterpret_cast<float4 *>(dW + e) = acc;
  else if (colsum)
    *reinterpret_cast<float4 *>(colsum + (e - n_dw)) = acc;
}

struct WgradPlan {
  int S;
  int64_t rows_per_slice;
  int tj;
};
inline WgradPlan wgrad_plan(int64_t M, int Na, int Nb) {
  WgradPlan p;
  p.tj = Nb % 256 == 0 ? 256 : 128;
  const int64_t ntiles = (int64_t)(Na / 128) * (Nb / p.tj);
  const int64_t nstages = (M + kRS - 1) / kRS;
  int64_t S = 512 / ntiles;                    // about two workgroups per CU in one round
  if (S < 1) S = 1;
  if (S > nstages) S = nstages > 0 ? nstages : 1;
  p.rows_per_slice = ((nstages + S - 1) / S) * kRS;
  p.S = (int)((M + p.rows_per_slice - 1) / p.rows_per_slice);
  if (p.S < 1) p.S = 1;
  return p;
}

}  // namespace

extern "C" size_t ampconv_proj_weight_image_bytes(int N, int K) {
  if (N <= 0 || K <= 0) return 0;
  return (size_t)N * (size_t)K * 6;
}

extern "C" int ampconv_proj_supported(int N, int K) {
  return N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0;
}

extern "C" int ampconv_proj_weight_image(const float *W, int64_t stride_n, int64_t stride_k, int N, int K,
                                         void *image, void *stream) {
  if (!ampconv_proj_supported(N, K)) return AMPCONV_E_BADARG;
  if (!W || !image || (uintptr_t)image % 16) return AMPCONV_E_BADARG;
  const int total = N * (K / 8);
  weight_image_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(W, stride_n, stride_k, N, K,
                                                                            (char *)image);
  return ampconv_launch_status();
}

extern "C" int ampconv_proj_rows(const float *A, int64_t lda, int64_t M, int K, const void *wimage, int N,
                                 const float *bias, const int32_t *rowptr, int L, float *out, int64_t ldc,
                                 void *stream) {
  if (M < 0 || !ampconv_proj_supported(N, K) || lda < K || ldc < N || lda % 4) return AMPCONV_E_BADARG;
  if (M == 0) return AMPCONV_OK;
  if (!A || !wimage || !out || (uintptr_t)A % 16 || (uintptr_t)wimage % 16) return AMPCONV_E_BADARG;
  if (rowptr && L <= 0) return AMPCONV_E_BADARG;
  const int64_t rts = (M + kBM - 1) / kBM;
  if (rts > (int64_t)INT32_MAX / 64) return AMPCONV_E_BADARG;
  RowsArgs a{A, lda, M, K, N, (const char *)wimage, bias, rowptr, L, out, ldc, (int)rts};
  const int64_t rtp = (rts + 7) / 8 * 8;
  const int nct = N / 128;
  proj_rows_kernel<128><<<(unsigned)(rtp * nct), 256, 0, (hipStream_t)stream>>>(a);
  return ampconv_launch_status();
}

extern "C" size_t ampconv_proj_wgrad_workspace_bytes(int64_t M, int Na, int Nb) {
  if (M < 0 || Na <= 0 || Nb <= 0 || Na % 128 || Nb % 128) return 0;
  const WgradPlan p = wgrad_plan(M, Na, Nb);
  return (size_t)p.S * ((size_t)Na * Nb + Na) * sizeof(float);
}

extern "C" int ampconv_proj_wgrad(const float *A, int64_t lda, const float *B, int64_t ldb, int64_t M, int Na,
                                  int Nb, const int32_t *rowptr, int L, float *dW, float *colsum,
                                  void *workspace, size_t workspace_bytes, void *stream) {
  if (M < 0 || Na <= 0 || Nb <= 0 || Na % 128 || Nb % 128 || lda < Na || ldb < Nb || lda % 4 || ldb % 4)
    return AMPCONV_E_BADARG;
  if (!dW || (uintptr_t)dW % 16 || (colsum && (uintptr_t)colsum % 16)) return AMPCONV_E_BADARG;
  if (rowptr && L <= 0) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    hipError_t e = hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Na * Nb, st);
    if (e == hipSuccess && colsum) e = hipMemsetAsync(colsum, 0, sizeof(float) * Na, st);
    return e == hipSuccess ? AMPCONV_OK : (int)e;
  }
  if (!A || !B || (uintptr_t)A % 16 || (uintptr_t)B % 16 || !workspace || (uintptr_t)workspace % 16)
    return AMPCONV_E_BADARG;
  const WgradPlan p = wgrad_plan(M, Na, Nb);
  const size_t n_all = (size_t)Na * Nb + Na;
  if (workspace_bytes < (size_t)p.S * n_all * sizeof(float)) return AMPCONV_E_WORKSPACE;
  WgradArgs a{A, lda, B, ldb, M, Na, Nb, rowptr, L, (float *)workspace, p.S, p.rows_per_slice};
  const int ntiles = (Na / 128) * (Nb / p.tj);
  const unsigned grid = (unsigned)(((p.S + 7) / 8 * 8) * ntiles);
  if (p.tj == 256) {
    if (rowptr) proj_wgrad_kernel<256, true><<<grid, 256, 0, st>>>(a);
    else proj_wgrad_kernel<256, false><<<grid, 256, 0, st>>>(a);
  } else {
    if (rowptr) proj_wgrad_kernel<128, true><<<grid, 256, 0, st>>>(a);
    else proj_wgrad_kernel<128, false><<<grid, 256, 0, st>>>(a);
  }
  wgrad_reduce_kernel<<<(unsigned)((n_all / 4 + 255) / 256), 256, 0, st>>>((const float *)workspace, p.S,
                                                                          (int64_t)Na * Nb, (int64_t)n_all, dW, colsum);
  return ampconv_launch_status();
}
