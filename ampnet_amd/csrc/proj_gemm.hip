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
// SCALED MODE (the caller passes the largest finite magnitude of each fp32 operand, ampconv_absmax / proj_rows'
// out_absmax): x' = x * 2^(14 - floor(log2 max)) fits fp16's range, x' = h1 + h2 with h1 = fp16(x'), h2 = fp16(x' - h1)
// (2 x 11 bits + the sign of the remainder: |x' - h1 - h2| <= 2^-22 |x'|; an element below 2^-17 of the tensor's
// maximum has a subnormal remainder, absolute error <= 2^-25 in scaled units = 2^-39 of the maximum), and a product is
// the fp32 sum of THREE v_mfma_f32_32x32x16_f16 partial products a1 b1 + a1 b2 + a2 b1 (dropped: a2 b2 <= 2^-22 |a b|);
// the power-of-two scales leave through the epilogue, exactly.  Measured against fp64 the result is as close as the
// six-product one (half the accumulator roundings) and closer than the library's fp32 GEMM; half the
// matrix instructions, two thirds of the split and LDS work.
//
// Two kernels:
//   proj_rows  out[M, N] = A[M, K] W^T (+ bias) (* row mask)      A fp32 rows, split on the way into LDS;
//              W pre-split ONCE into an image of ready MFMA fragments (ampconv_proj_weight_image) that is
//              copied to LDS by LDS-DMA (no registers, no address arithmetic, conflict-free by construction)
//   proj_wgrad dW[Na, Nb] = A[M, Na]^T B[M, Nb], colsum(A)         (weight / bias gradients: reduction over
//              the N*L node-token rows; both operands split on the fly, fragments by the hardware-transposing
//              ds_read_b64_tr_b16; deterministic two-step reduction over row slices)
#include <stdlib.h>
#include <type_traits>
#include "proj_common.h"

namespace {
using namespace proj;

// HP = false: three bf16 planes, six products (exact split, any range); true: two fp16 planes of the SCALED operand,
// three products
template <bool HP>
struct Pl {
  static constexpr int NP = HP ? 2 : 3;           // planes
  static constexpr int NQ = HP ? 3 : 6;           // partial products, smallest first: planes (PA[q], PB[q])
  static constexpr int kTile = NP * kFrag;        // the planes of one (32-row tile, 16-deep k step)
};
__device__ constexpr int kPA[2][6] = {{2, 1, 0, 1, 0, 0}, {1, 0, 0, 0, 0, 0}};
__device__ constexpr int kPB[2][6] = {{0, 1, 2, 0, 1, 0}, {0, 1, 0, 0, 0, 0}};
template <bool HP>
__device__ __forceinline__ f32x16 mfma_p(const i32x4 &a, const i32x4 &b, const f32x16 &c) {
  if constexpr (HP) return MFMA32H(a, b, c);
  else return MFMA32(a, b, c);
}

// scale of a tensor whose largest finite magnitude is `amax`: 2^(14 - floor(log2 amax)), so that the scaled maximum
// lies in [2^14, 2^15) (fp16: finite below 2^16); exponent field clamped to [15, 254] (zero / subnormal maxima take
// 2^126, a non-finite one -- never produced by ampconv_absmax -- 2^-113)
__device__ __forceinline__ float plane_scale(float amax) {
  int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xFFu);
  e = e < 15 ? 15 : (e > 254 ? 254 : e);
  return __builtin_bit_cast(float, (unsigned)(268 - e) << 23);
}
__device__ __forceinline__ float plane_unscale(float amax) { return 1.f / plane_scale(amax); }     // exact: a power of two

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

// (x0, x1) * s -> packed fp16 pairs of the two planes (h3 unused)
__device__ __forceinline__ Pair3 split_pair_h(float x0, float x1, float s) {
  x0 *= s;
  x1 *= s;
  const unsigned a = cvt_pk_f16(x0, x1);
  const f16x2 av = __builtin_bit_cast(f16x2, a);
  const float r0 = x0 - (float)av[0], r1 = x1 - (float)av[1];
  return Pair3{(int)a, (int)cvt_pk_f16(r0, r1), 0};
}
template <bool HP>
__device__ __forceinline__ Pair3 split_pair_t(float x0, float x1, float s) {
  if constexpr (HP) return split_pair_h(x0, x1, s);
  else return split_pair(x0, x1);
}

// the partial products for a ROW of accumulators that share the A fragment, plane-pair major: consecutive MFMAs go to
// different accumulators (a dependent 32x32x16 waits ~25 % of its own length for the previous result: K = 768 row
// product 40.7 -> 38.3 ms); every accumulator receives its products smallest first
template <bool HP, int NJ>
__device__ __forceinline__ void mfma_row(const i32x4 (&a)[3], const i32x4 (&b)[NJ][3], f32x16 (&c)[NJ]) {
#pragma unroll
  for (int q = 0; q < Pl<HP>::NQ; ++q)
#pragma unroll
    for (int j = 0; j < NJ; ++j) c[j] = mfma_p<HP>(a[kPA[HP][q]], b[j][kPB[HP][q]], c[j]);
}

#ifdef AMPCONV_PROJ_STAMPS
// diagnostic build (tools/stamp_proj.py): per-phase s_memtime sums of proj_rows_kernel, written to a device
// buffer that nothing else reads; never part of the product build
__device__ unsigned long long g_proj_stamps[8 * 4096];
#define PSTAMP_DECL unsigned long long st_last, st_now, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#define PSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory"); \
  __builtin_amdgcn_sched_barrier(0); st_acc[i] += st_now - st_last; st_last = st_now; } while (0)
#define PSTAMP_FLUSH(unit) do { if ((unit) < 4096 && threadIdx.x == 0) for (int i_ = 0; i_ < 8; ++i_) \
  g_proj_stamps[(unit) * 8 + i_] = st_acc[i_]; } while (0)
#else
#define PSTAMP_DECL
#define PSTAMP(i)
#define PSTAMP_FLUSH(unit)
#endif

// ---------------------------------------------------------------------------------------------------
// weight image: for k step ks (16 deep), 32-column tile n32 of the OUTPUT, plane p:
//   fragment (ks * N/32 + n32) * NP + p,  1 KiB, lane l = (r = l & 31, h = l >> 5) holds plane p of
//   B[n32 * 32 + r][ks * 16 + 8 h + 0..7]      with B[n][k] = W[n * stride_n + k * stride_k]
// i.e. exactly the B operand of v_mfma_f32_32x32x16_{bf16,f16}, in lane order; the fragments of one k step and one
// block of columns are contiguous, so a stage is one linear LDS-DMA copy.
// An fp32 image holds BOTH forms: [three bf16 planes: Np Kp 6 bytes][two fp16 planes of the scaled weight: Np Kp 4 bytes]
// [512 bytes: 64 partial maxima and, at [64], the weight's largest finite magnitude, which the scaled kernels read back]
struct ImageJob {
  const float *W;
  int64_t sn, sk;
  int N, K;
  char *img;
};
struct ImageJobs {
  ImageJob j[8];
};
__host__ __device__ inline size_t image_half_offset(int N, int K) {
  return (size_t)((N + 127) / 128 * 128) * (size_t)((K + 31) / 32 * 32) * 6;
}
__host__ __device__ inline size_t image_amax_offset(int N, int K) {
  return (size_t)((N + 127) / 128 * 128) * (size_t)((K + 31) / 32 * 32) * 10;
}
// 64 workgroups per weight, each the largest finite magnitude of its share, read in memory order; no atomics (nothing to
// zero first): the partial maxima go to the image's trailer [64 floats], weight_image_kernel folds them ([64] = the result)
constexpr int kAmaxParts = 64;
__global__ __launch_bounds__(256) void weight_absmax_kernel(ImageJobs jobs) {
  const ImageJob jb = jobs.j[blockIdx.y];
  __shared__ float red[4];
  const bool kfast = jb.sk == 1;                              // which index walks memory fastest
  const int inner = kfast ? jb.K : jb.N;
  const int64_t si = kfast ? jb.sn : jb.sk;                   // stride of the slow index
  const int64_t total = (int64_t)jb.N * jb.K;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)kAmaxParts * 256) {
    const int64_t o = i / inner;
    m = fmaxf(m, finite_abs(jb.W[o * si + (i - o * inner) * (kfast ? jb.sk : jb.sn)]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    reinterpret_cast<float *>(jb.img + image_amax_offset(jb.N, jb.K))[blockIdx.x] =
        fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__device__ __forceinline__ float folded_amax(const char *trailer) {
  const float4 *p = reinterpret_cast<const float4 *>(trailer);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < kAmaxParts / 4; ++i) {
    const float4 v = p[i];
    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  return m;
}
__global__ void weight_image_kernel(ImageJobs jobs) {
  const ImageJob jb = jobs.j[blockIdx.y];
  // the image covers the PADDED shape (columns to a multiple of 128, depth to a multiple of 32): zeros beyond N and K
  const int Np = (jb.N + 127) / 128 * 128, Kp = (jb.K + 31) / 32 * 32;
  const int k8s = Kp / 8;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Np * k8s) return;
  const int n = idx / k8s, k8 = idx - n * k8s;
  const float wmax = folded_amax(jb.img + image_amax_offset(jb.N, jb.K));
  if (idx == 0) reinterpret_cast<float *>(jb.img + image_amax_offset(jb.N, jb.K))[kAmaxParts] = wmax;
  const float sw = plane_scale(wmax);
  float x[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    x[j] = (n < jb.N && 8 * k8 + j < jb.K) ? jb.W[(int64_t)n * jb.sn + (int64_t)(8 * k8 + j) * jb.sk] : 0.f;
  i32x4 pl[3], ph[2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const Pair3 q = split_pair(x[2 * t], x[2 * t + 1]);
    pl[0][t] = q.h1; pl[1][t] = q.h2; pl[2][t] = q.h3;
    const Pair3 qh = split_pair_h(x[2 * t], x[2 * t + 1], sw);
    ph[0][t] = qh.h1; ph[1][t] = qh.h2;
  }
  const int ks = k8 >> 1, h = k8 & 1, n32 = n >> 5, r = n & 31;
  char *dst = jb.img + ((size_t)(ks * (Np / 32) + n32) * 3) * kFrag + (32 * h + r) * 16;
#pragma unroll
  for (int p = 0; p < 3; ++p) *reinterpret_cast<i32x4 *>(dst + p * kFrag) = pl[p];
  dst = jb.img + image_half_offset(jb.N, jb.K) + ((size_t)(ks * (Np / 32) + n32) * 2) * kFrag + (32 * h + r) * 16;
#pragma unroll
  for (int p = 0; p < 2; ++p) *reinterpret_cast<i32x4 *>(dst + p * kFrag) = ph[p];
}

// ---------------------------------------------------------------------------------------------------
// proj_rows: BM x BN output tiles per workgroup of WM x WN waves (each (BM / WM) x (BN / WN)), K in steps of 16.
// LDS: two buffers of {weight fragments BN / 32 * 3 KiB, A planes BM / 32 * 3 KiB}; ONE barrier per step: while a
// step's fragments are multiplied, the next step's weight fragments arrive by LDS-DMA and the next step's rows
// (loaded as whole 128-byte lines, 32 deep, one step ahead of their first use) are split and filed into the other
// buffer, so a wave's vector / LDS / memory instructions sit in the shadow of its own and its SIMD partner's MFMAs.
// Workgroups are PERSISTENT: one flat step loop over a strided list of tiles, the first fragments of the next tile
// are on their way before the store tail of the current one.
// What the shape is chosen by (in-kernel stamps, tools/stamp_proj.py): every vector-memory instruction a wave
// issues (1-KiB row load, 1-KiB LDS-DMA piece, store) costs it ~100 cycles once all waves of a CU use the address
// path, against 32 per MFMA: a 128 x 128 tile of four waves needs 0.38 of them per MFMA, 128 x 256 needs 0.19.
// Output tiles leave through LDS as whole 128-byte row segments (dwordx4 stores: a quarter of the store
// instructions of the C/D layout).
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
  int row_tiles;              // ceil(M / BM)
  int64_t tiles;              // tile slots: row tiles rounded up to 8, times column tiles
  int Kp, Np;                 // K, N padded to multiples of 32 / 128 (= the weight image's shape)
  // scaled mode (HP): largest finite magnitudes of A and of the weight (device floats); the output's may be recorded
  const float *amax_a, *amax_w;
  float *out_amax;            // or null: atomic max over this launch's outputs (the caller zeroes it), scaled mode only
  int amax_col0;              // ... over the output columns >= amax_col0 only (the V third of a packed in-projection)
  // PLANES (scaled mode only): `out` receives, per 32-column block, the two fp16 planes of value * plane_scale(*out_bound)
  // (the format of csrc/edge_mfma_f16x2.hip); out_bound: device float, an upper bound of the output's magnitudes
  const float *out_bound;
  int plane_dh;               // ... channels per slot: 32 (128-byte slots) or 16 (64-byte slots: two heads per 32-column block)
  int row_scale;              // with rowptr: 0 = rows of nodes with an empty segment come out 0 (the mask),
                              // 1 = every row is DIVIDED by its node's segment length as well (1 / in-degree)
};

// RAGGED: K % 32 != 0 or N % BN != 0 (e.g. the reference's default embed_dim = 100): row loads beyond K read as zero,
// columns beyond N are computed on the image's zero padding and not stored
#ifndef AMPCONV_PROJ_ROWS_128_WAVES
#define AMPCONV_PROJ_ROWS_128_WAVES 2      // (A/B: 3 = the 128 x 128 scaled shapes compiled for three workgroups per CU)
#endif
template <int BM, int BN, int WM, int WN, bool RAGGED, bool HP, bool PLANES = false>
__global__ __launch_bounds__(64 * WM * WN, (BM == 128 && BN == 128 && HP) ? AMPCONV_PROJ_ROWS_128_WAVES : 2) void proj_rows_kernel(RowsArgs a) {
  static_assert(!PLANES || HP, "plane output exists in the scaled mode only");
  constexpr int kNP = Pl<HP>::NP, kTile3 = Pl<HP>::kTile;
  constexpr int NW = WM * WN, NTHR = 64 * NW;
  constexpr int MTB = BM / 32, NTB = BN / 32;          // 32-row / 32-column tiles per workgroup
  constexpr int MTW = MTB / WM, NTW = NTB / WN;        // ... per wave
  constexpr int kStepA = MTB * kTile3, kStepB = NTB * kTile3;
  constexpr int kBuf = kStepB + kStepA;                // one buffer: [weight fragments][A planes]
  constexpr int kPieces = kStepB / kFrag;              // 1-KiB DMA pieces per step
  constexpr int NLH = BM * 4 / NTHR;                   // float4 row loads per thread and 16-deep half line
  constexpr int kRowsPerLoad = NTHR / 4;
  static_assert(kPieces % NW == 0 && NLH * NTHR == BM * 4 && kRowsPerLoad % 32 == 0, "tile / wave shape");
  static_assert(NW * 4096 <= kBuf, "store staging lives in the consumed buffer");
  // one LDS object: [buffer 0][buffer 1][row flags of the even / odd tile]
  __shared__ __attribute__((aligned(16))) char smem[2 * kBuf + 2 * BM * 4];
  float *flags = reinterpret_cast<float *>(smem + 2 * kBuf);

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = w / WN, wn = w % WN;
  const int nct = (a.Np + BN - 1) / BN;

  // rows of this thread: float4 column c of a 16-deep half line, rows r0 + kRowsPerLoad * i
  const int c = t & 3, r0 = t >> 2;
  // where its split chunks go: fragment (32-row tile, plane), slot 32 h + (r ^ 4 h), half q  (conflict-free
  // 8-byte stores and 16-byte fragment reads: tools/lds_layout_check.py, entry proj)
  const int wh = c >> 1, wq = c & 1;
  const int wdst = kStepB + (r0 >> 5) * kTile3 + ((32 * wh + ((r0 & 31) ^ (4 * wh))) << 4) + 8 * wq;
  constexpr int kWStep = (kRowsPerLoad / 32) * kTile3;         // LDS bytes between the tiles of loads i, i + 1
  // fragment reads: lane (r = lane & 31, h = lane >> 5)
  const int fr = lane & 31, fh = lane >> 5;
  const int ard = kStepB + (MTW * wm) * kTile3 + ((32 * fh + (fr ^ (4 * fh))) << 4);
  const int brd = (NTW * wn) * kTile3 + lane * 16;
  const size_t wstep = (size_t)(a.Np / 32) * kTile3;           // bytes of one k step of the whole image
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_char *)smem;
  // scaled mode: operand scale, and what undoes it and the weight's in the epilogue (two exact factors: their product
  // could leave the fp32 range)
  const float sa = HP ? plane_scale(*a.amax_a) : 1.f;
  const float ua = HP ? 1.f / sa : 1.f, uw = HP ? plane_unscale(*a.amax_w) : 1.f;
  const float so = PLANES ? plane_scale(*a.out_bound) : 1.f;      // PLANES: the output leaves in these units
  float omax = 0.f;                     // largest finite |output| this thread has stored

  // tile slot u -> (row tile, column tile).  Workgroups b, b + 8, ... share an XCD (round-robin dispatch, speed
  // only): consecutive slots of one XCD are the column tiles of ONE row tile, whose rows then come from its L2
  struct Tile {
    int64_t row0;
    int col0;
    bool valid;
  };
  auto tile_of = [&](int64_t u) {
    Tile tl;
    const int64_t i_x = u / kXcd;
    const int64_t rt = (i_x / nct) * kXcd + u % kXcd;
    tl.row0 = rt * BM;
    tl.col0 = (int)(i_x % nct) * BN;
    tl.valid = u < a.tiles && rt < a.row_tiles;
    return tl;
  };
  auto dma_step = [&](const Tile &tl, int ks, int buf) {
    const char *src = a.wimg + (size_t)(tl.col0 / 32) * kTile3 + (size_t)ks * wstep + lane * 16;
    const unsigned dst = lds0 + buf * kBuf;
#pragma unroll
    for (int j = 0; j < kPieces / NW; ++j) {
      const int piece = w + NW * j;
      dma16(src + piece * kFrag, dst + piece * kFrag);
    }
  };
  // rows of the thread in the NEXT line block to request (advanced 32 floats per block, re-based at a tile seam)
  const float *rp[NLH];
  int kpos = 4 * c;                     // RAGGED: column of the thread's first float4 in the block to request
  auto rebase = [&](const Tile &tl) {
    kpos = 4 * c;
#pragma unroll
    for (int i = 0; i < NLH; ++i) {
      int64_t m = tl.row0 + kRowsPerLoad * i + r0;
      m = m < a.M ? m : a.M - 1;
      rp[i] = a.A + m * a.lda + 4 * c;
    }
  };
  // Row loads are inline assembly, their waits counted by hand: hipcc's wait bookkeeping does not see the LDS-DMA
  // pieces (also assembly), so the wait it would put in front of the first use of an ordinary load -- "all my loads but
  // the n youngest" -- also drains every DMA piece issued since: a full L2 round trip in the even step, in plain sight
  // once the products per step were halved (in-kernel stamps: even step 3306 cycles, odd step 1885).
  // (the two ragged six-product shapes -- the reference's embed_dim = 100, no benchmark shape -- sit at the register
  // limit and spill an address pair; a spill between an assembly load and its wait would save a register whose data has
  // not landed, so THEY keep compiler-managed loads: hipcc then waits by itself wherever it touches the values, at the
  // price of the drained DMA pieces described above)
  constexpr bool kAsmRows = !(RAGGED && !HP);
  f32x4 x[2][NLH];                      // one 32-deep line block of the thread's rows: [half][row]
  auto gload = [](const float *p) {
    f32x4 v;
    if constexpr (kAsmRows) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    else v = *reinterpret_cast<const f32x4 *>(p);
    return v;
  };
// behind a wait: the values of half H are used after this point only
#define ROWS_LANDED(H)                                                   \
  do {                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < NLH; ++i_) asm volatile("" : "+v"(x[H][i_])); \
  } while (0)
  auto load_rows = [&]() {
#pragma unroll
    for (int i = 0; i < NLH; ++i) {
      if (!RAGGED) {
        x[0][i] = gload(rp[i]);
        x[1][i] = gload(rp[i] + 16);
      } else {
        // The even step waits for the weight DMAs with a COUNTED vmcnt (2 * NLH younger row loads stay in flight), so
        // the number of vector-memory instructions a wave issues here must not depend on K: a conditional load is
        // skipped by the whole wave when no lane needs it (K % 32 in [1, 16]: the reference's embed_dim = 100) and the
        // count would then cover this wave's DMA pieces only by luck.  Load unconditionally from an in-bounds address
        // (the row's first float4 where the column lies beyond K) and zero by select.
        const bool in0 = kpos < a.K, in1 = kpos + 16 < a.K;
        // (zeroed by select where the VALUES are first used, split_rows: nothing may touch the registers before the wait)
        x[0][i] = gload(in0 ? rp[i] : rp[i] - kpos);
        x[1][i] = gload(in1 ? rp[i] + 16 : rp[i] - kpos);
      }
      rp[i] += 32;
    }
    kpos += 32;
  };
  // (kcol: RAGGED only, the column of the thread's float4 in the block being split: beyond K it reads as zero)
  auto split_rows = [&](auto half, int buf, int kcol) {       // `half` is a compile-time constant: x stays in registers
    char *base = smem + buf * kBuf + wdst;
    const bool inb = !RAGGED || kcol + 16 * decltype(half)::value < a.K;
#pragma unroll
    for (int i = 0; i < NLH; ++i) {
      f32x4 v = x[decltype(half)::value][i];
      if (RAGGED) v = inb ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      const Pair3 p0 = split_pair_t<HP>(v.x, v.y, sa), p1 = split_pair_t<HP>(v.z, v.w, sa);
      char *d = base + i * kWStep;
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kFrag) = i32x2{p0.h2, p1.h2};
      if (kNP == 3) *reinterpret_cast<i32x2 *>(d + 2 * kFrag) = i32x2{p0.h3, p1.h3};
    }
  };
  f32x16 acc[MTW][NTW];
  auto compute = [&](int buf) {
    const char *bb = smem + buf * kBuf;
    i32x4 bf[NTW][3];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
      for (int p = 0; p < kNP; ++p) bf[j][p] = *reinterpret_cast<const i32x4 *>(bb + brd + j * kTile3 + p * kFrag);
    // accumulator rows in groups of G so that consecutive MFMAs are at least four accumulators apart (mfma6_row)
    constexpr int G = NTW >= 4 ? 1 : (4 / NTW < MTW ? 4 / NTW : MTW);
    static_assert(MTW % G == 0, "accumulator rows are taken in groups");
#pragma unroll
    for (int i = 0; i < MTW; i += G) {
      i32x4 af[G][3];
#pragma unroll
      for (int ii = 0; ii < G; ++ii)
#pragma unroll
        for (int p = 0; p < kNP; ++p) af[ii][p] = *reinterpret_cast<const i32x4 *>(bb + ard + (i + ii) * kTile3 + p * kFrag);
      if constexpr (G == 1) {
        mfma_row<HP, NTW>(af[0], bf, acc[i]);
      } else {
#pragma unroll
        for (int q = 0; q < Pl<HP>::NQ; ++q)
#pragma unroll
          for (int ii = 0; ii < G; ++ii)
#pragma unroll
            for (int j = 0; j < NTW; ++j)
              acc[i + ii][j] = mfma_p<HP>(af[ii][kPA[HP][q]], bf[j][kPB[HP][q]], acc[i + ii][j]);
      }
    }
  };

  PSTAMP_DECL
  int64_t u = blockIdx.x;
  Tile cur = tile_of(u);
  if (!cur.valid) return;        // slots are ordered: nothing further for this workgroup either
  Tile nxt = tile_of(u + gridDim.x);
  int par = 0;                   // parity of the tile (row flags)
  const int KB = a.Kp / 32;      // line blocks per tile; a tile is 2 KB steps, so step parity = buffer, statically
  rebase(cur);
  dma_step(cur, 0, 0);
  load_rows();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ROWS_LANDED(0);
  ROWS_LANDED(1);
  split_rows(std::integral_constant<int, 0>{}, 0, kpos - 32);

  for (;;) {
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (t < BM) {
      float f = 1.f;
      if (a.rowptr) {
        const int64_t m = cur.row0 + t < a.M ? cur.row0 + t : a.M - 1;
        const unsigned node = (unsigned)m / (unsigned)a.L;        // (M < 2^31 with a mask: host check)
        const int seg = a.rowptr[node + 1] - a.rowptr[node];
        f = seg != 0 ? (a.row_scale ? 1.f / (float)seg : 1.f) : 0.f;
      }
      flags[par * BM + t] = f;
    }
    for (int kb = 0; kb < KB; ++kb) {
      // ---- even step (fragments in buffer 0): the odd step's weight fragments by DMA, its rows = the second half
      // of the line block in registers; then the block is used up and the next one is requested
      PSTAMP(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // buffer 0 complete, buffer 1 free
      PSTAMP(1);
      dma_step(cur, 2 * kb + 1, 1);
      compute(0);
      split_rows(std::integral_constant<int, 1>{}, 1, kpos - 32);       // (its half landed with the last full wait)
      const bool last = kb + 1 == KB;
      if (last) rebase(nxt);
      const bool request = !last || nxt.valid;      // workgroup-uniform
      if (request) {
        load_rows();
        // the DMA pieces have had the compute phase to land; the row loads requested after them stay in flight
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLH) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      PSTAMP(2);
      // ---- odd step (buffer 1): next even step's fragments (this tile's, or step 0 of the next tile; past the
      // end of the work list the refill is harmless and keeps the step free of branches)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PSTAMP(3);
      dma_step(last ? (nxt.valid ? nxt : cur) : cur, last ? 0 : 2 * kb + 2, 0);
      compute(1);
      // the row block requested in the even step (older than this step's DMA pieces, which stay in flight)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces / NW) : "memory");
      ROWS_LANDED(0);
      split_rows(std::integral_constant<int, 0>{}, 0, kpos - 32);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ROWS_LANDED(1);
      PSTAMP(4);
    }
    // store tail.  C/D register e of lane (col = lane & 31, hi = lane >> 5) is row (e & 3) + 8 (e >> 2) + 4 hi of a
    // 32 x 32 tile: bias and row mask applied in that layout, then the tile goes through a wave-private 4 KiB of
    // LDS -- inside buffer 1, which the last step consumed: free once every wave has passed the barrier below, and
    // not refilled before the next even step's barrier -- and leaves as 8 rows x 128 bytes per store instruction.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      const float *fl_t = flags + par * BM;
      float *stage = reinterpret_cast<float *>(smem + kBuf + w * 4096);
      const int sr = lane >> 3, sc4 = lane & 7;
      // the wave's bias entries, ALL before the first store: a load behind a store waits for that store's
      // acknowledgement as well (vmcnt counts in issue order) -- loaded per sub-tile, each of the eight sub-tiles paid a
      // store round trip plus an L2 round trip
      float bj[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int col = cur.col0 + (NTW * wn + j) * 32 + fr;
        bj[j] = (a.bias && (!RAGGED || col < a.N)) ? a.bias[col] : 0.f;
      }
      // PLANES: the 32 x 32 sub-tile is one (row, head) slot per row -- 32 fp16 hi then 32 fp16 lo of value * so.  The
      // staged values are already in plane units (both operand scales and so folded into one factor, exact powers of
      // two).  Staging swaps the two 4-float chunks of every chunk pair in the rows with bit 1 set, so that lane
      // (row = lane >> 2, c = lane & 3) reads its 8 consecutive channels 8 c .. 8 c + 7 as two conflict-free
      // ds_read_b128 (rows r, r + 1 take the even physical chunks, r + 2, r + 3 the odd ones); it splits them and
      // stores 16 bytes of each plane: four dwordx4 stores per sub-tile, as the fp32 tail.
      auto store_planes = [&](auto ragged) {
        const float kso = ua * uw * so;
        const int pr = lane >> 2, pc = lane & 3, ps = (pr >> 1) & 1;
        // byte offsets of this lane's 8 channels in the 128 bytes of the sub-tile row: slots of 32 channels hold
        // [32 hi | 32 lo], slots of 16 channels [16 hi | 16 lo] twice
        const int ohi = a.plane_dh == 32 ? 16 * pc : 64 * (pc >> 1) + 16 * (pc & 1), olo = ohi + 2 * a.plane_dh;
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int rl0 = (MTW * wm + i) * 32;
          float fl[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 f4 = *reinterpret_cast<const float4 *>(fl_t + rl0 + 4 * fh + 8 * g);
            fl[4 * g] = f4.x; fl[4 * g + 1] = f4.y; fl[4 * g + 2] = f4.z; fl[4 * g + 3] = f4.w;
          }
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const int colt = cur.col0 + (NTW * wn + j) * 32;
            const float bjs = bj[j] * so;
#pragma unroll
            for (int e = 0; e < 16; ++e)
              stage[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + (fr ^ (((e >> 1) & 1) << 2))] = (acc[i][j][e] * kso + bjs) * fl[e];
            const bool rec = a.out_amax && colt >= a.amax_col0;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const int row = pr + 16 * g;
              const float4 v0 = *reinterpret_cast<const float4 *>(stage + row * 32 + 4 * ((2 * pc) ^ ps));
              const float4 v1 = *reinterpret_cast<const float4 *>(stage + row * 32 + 4 * ((2 * pc + 1) ^ ps));
              const Pair3 p0 = split_pair_h(v0.x, v0.y, 1.f), p1 = split_pair_h(v0.z, v0.w, 1.f);
              const Pair3 p2 = split_pair_h(v1.x, v1.y, 1.f), p3 = split_pair_h(v1.z, v1.w, 1.f);
              if ((!decltype(ragged)::value || cur.row0 + rl0 + row < a.M) && (!RAGGED || colt < a.N)) {
                char *o = reinterpret_cast<char *>(a.out + (cur.row0 + rl0 + row) * a.ldc + colt);
                *reinterpret_cast<i32x4 *>(o + ohi) = i32x4{p0.h1, p1.h1, p2.h1, p3.h1};
                *reinterpret_cast<i32x4 *>(o + olo) = i32x4{p0.h2, p1.h2, p2.h2, p3.h2};
                if (rec) omax = finite_abs_max(finite_abs_max(omax, v0), v1);       // (in plane units: undone at the end)
              }
            }
          }
        }
      };
      auto store_tile = [&](auto ragged) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
          const int rl0 = (MTW * wm + i) * 32;
          float fl[16];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 f4 = *reinterpret_cast<const float4 *>(fl_t + rl0 + 4 * fh + 8 * g);
            fl[4 * g] = f4.x; fl[4 * g + 1] = f4.y; fl[4 * g + 2] = f4.z; fl[4 * g + 3] = f4.w;
          }
#pragma unroll
          for (int j = 0; j < NTW; ++j) {
            const int colt = cur.col0 + (NTW * wn + j) * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e)
              stage[((e & 3) + 8 * (e >> 2) + 4 * fh) * 32 + fr] =
                  ((HP ? acc[i][j][e] * ua * uw : acc[i][j][e]) + bj[j]) * fl[e];
            float *o = a.out + (cur.row0 + rl0 + sr) * a.ldc + colt + 4 * sc4;
            const bool rec = HP && a.out_amax && colt >= a.amax_col0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 v = *reinterpret_cast<const float4 *>(stage + (sr + 8 * g) * 32 + 4 * sc4);
              if ((!decltype(ragged)::value || cur.row0 + rl0 + sr + 8 * g < a.M) && (!RAGGED || colt + 4 * sc4 < a.N)) {
                *reinterpret_cast<float4 *>(o + (int64_t)(8 * g) * a.ldc) = v;
                if (rec) omax = finite_abs_max(omax, v);
              }
            }
          }
        }
      };
      if constexpr (PLANES) {
        if (cur.row0 + BM <= a.M) store_planes(std::false_type{});
        else store_planes(std::true_type{});
      } else {
        if (cur.row0 + BM <= a.M)            // workgroup-uniform: only the last row tile is ragged
          store_tile(std::false_type{});
        else
          store_tile(std::true_type{});
      }
    }
    PSTAMP(6);
    if (!nxt.valid) break;
    cur = nxt;
    u += gridDim.x;
    nxt = tile_of(u + gridDim.x);
    par ^= 1;
  }
  if (HP && a.out_amax) {                 // one atomic per wave and launch; non-negative floats order as their bits
    if (PLANES) omax *= 1.f / so;         // recorded in plane units
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) omax = fmaxf(omax, __shfl_xor(omax, o));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(a.out_amax), __builtin_bit_cast(unsigned, omax));
  }
  PSTAMP_FLUSH(blockIdx.x);
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
  const float *rowflag;       // MASK: 1.0 / 0.0 per row (node of the row has an in-edge), in the workspace behind the slabs
  float *part;                // [S][Nap * Nbp + Nap]
  int S;
  int64_t rows_per_slice;     // multiple of 16
  int Nap, Nbp;               // Na, Nb padded to multiples of 128: the shape the tiles and the partial slabs cover
  const float *amax_a, *amax_b;   // scaled mode (HP): largest finite magnitudes of A and B (device floats)
};

constexpr int kRS = 16;       // rows per stage

// RAGGED: Na or Nb not a multiple of the tile (the reference's default embed_dim = 100): columns beyond them load as 0
template <int TI, int TJ, int WI, int WJ, bool MASK, bool RAGGED, bool HP>
__global__ __launch_bounds__(64 * WI * WJ, 2) void proj_wgrad_kernel(WgradArgs a) {
  constexpr int kNP = Pl<HP>::NP;
  constexpr int NW = WI * WJ, NTHR = 64 * NW;
  constexpr int kRowA = TI * 2, kRowB = TJ * 2;             // bytes per image row
  constexpr int kPlaneA = kRS * kRowA, kPlaneB = kRS * kRowB;
  constexpr int kStage = kNP * (kPlaneA + kPlaneB);
  constexpr int NIW = TI / 32 / WI, NJW = TJ / 32 / WJ;     // 32-column tiles of A / B per wave
  constexpr int kColsA4 = TI / 4, kColsB4 = TJ / 4;         // float4 per tile row
  constexpr int kRowsA = NTHR / kColsA4, kRowsB = NTHR / kColsB4;   // rows covered by one load of the workgroup
  constexpr int NLA = kRS / kRowsA, NLB = kRS / kRowsB;     // float4 loads per thread and stage
  static_assert(NLA * kRowsA == kRS && NLB * kRowsB == kRS && kRowsA * kColsA4 * 16 <= 2 * kStage, "tile shape");
  __shared__ __attribute__((aligned(16))) char smem[2 * kStage];

  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wi = w / WJ, wj = w % WJ;
  const int ntj = a.Nbp / TJ, ntiles = (a.Nap / TI) * ntj;
  const int b = blockIdx.x, xcd = b % kXcd, i_x = b / kXcd;
  const int slice = (i_x / ntiles) * kXcd + xcd, tile = i_x % ntiles;
  if (slice >= a.S) return;
  const int ti = tile / ntj, tj = tile % ntj;
  const int64_t m0 = (int64_t)slice * a.rows_per_slice;
  const int64_t m1 = m0 + a.rows_per_slice < a.M ? m0 + a.rows_per_slice : a.M;
  const int ns = (int)((m1 - m0 + kRS - 1) / kRS);

  // loads: A rows ra + kRowsA i (float4 column ca), B rows rb + kRowsB i (float4 column cb)
  const int ca = t % kColsA4, ra = t / kColsA4;
  const int cb = t % kColsB4, rb = t / kColsB4;
  const float *pa = a.A + (int64_t)ti * TI + 4 * ca;
  const float *pb = a.B + (int64_t)tj * TJ + 4 * cb;
  // plane stores: 8 bytes at row r, byte (8 c) ^ ((r & 3) << 6)
  int wa[NLA], wb[NLB];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int r = ra + kRowsA * i;
    wa[i] = r * kRowA + ((8 * ca) ^ ((r & 3) << 6));
  }
#pragma unroll
  for (int i = 0; i < NLB; ++i) {
    const int r = rb + kRowsB * i;
    wb[i] = kNP * kPlaneA + r * kRowB + ((8 * cb) ^ ((r & 3) << 6));
  }
  // transposed reads: lane = (h = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3) addresses
  // row 8 h + 4 u + q, columns 32 tile + 16 gi + 4 p .. + 3 (u = 0, 1: the two halves of the 8-deep k group)
  const int fh = lane >> 5, gi = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3, fr = lane & 31;
  const int rdA = (8 * fh + q) * kRowA + (q << 6) + 32 * gi + 8 * pp;       // ^ (tile & 3) << 6, + (tile >> 2) << 8
  const int rdB = kNP * kPlaneA + (8 * fh + q) * kRowB + (q << 6) + 32 * gi + 8 * pp;

  f32x16 acc[NIW][NJW];
#pragma unroll
  for (int i = 0; i < NIW; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  // scaled mode: the partial tiles are in scaled units (wgrad_reduce_kernel undoes both scales); the column sums are not
  const float sa = HP ? plane_scale(*a.amax_a) : 1.f, sb = HP ? plane_scale(*a.amax_b) : 1.f;

  // rows of TWO stages ahead live in registers (set = stage & 1): at three products per stage a stage lasts ~0.7 us, less
  // than a trip to HBM, and with one stage in flight the kernel waited for its loads (dWin 31.0 ms at cfg4's shape).  The
  // row flag (slice end / mask) is applied where the rows are split, not where they are requested: a multiply right
  // behind the load would wait for it there
  // (the 128 x 256 shape -- Na an odd multiple of 128 against Nb a multiple of 256, no layer shape of the benchmarks --
  // has no registers left for a second set: one stage ahead there)
  constexpr int NSET = (TI == 128 && TJ == 256) ? 1 : 2;
  float4 xa[NSET][NLA], xb[NSET][NLB];
  float fa[NSET][NLA];
  auto load_stage = [&](int s, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const int64_t mb = m0 + (int64_t)s * kRS;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      const int64_t m = mb + ra + kRowsA * i;
      const int64_t mc = m < m1 ? m : m1 - 1;
      if (!RAGGED || ti * TI + 4 * ca < a.Na) xa[set][i] = *reinterpret_cast<const float4 *>(pa + mc * a.lda);
      else xa[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      float f = m < m1 ? 1.f : 0.f;
      if (MASK) {
        // one flag per row, written by row_mask_kernel before this launch (the lanes of a row read one address; until the
        // end of round 4 every thread derived it here: row / L in 64 bits, two CSR bounds -- 280 instructions per stage)
        f *= a.rowflag[mc];
      }
      fa[set][i] = f;
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const int64_t m = mb + rb + kRowsB * i;
      const int64_t mc = m < m1 ? m : m1 - 1;
      if (!RAGGED || tj * TJ + 4 * cb < a.Nb) xb[set][i] = *reinterpret_cast<const float4 *>(pb + mc * a.ldb);
      else xb[set][i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_stage(0, std::integral_constant<int, 0>{});
  if (NSET == 2 && ns > 1) load_stage(1, std::integral_constant<int, NSET - 1>{});

  auto stage = [&](int s, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    char *buf = smem + (s & 1) * kStage;
#pragma unroll
    for (int i = 0; i < NLA; ++i) {
      float4 v = xa[set][i];
      if (MASK || s >= ns - 1) { const float f = fa[set][i]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; }
      cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
      const Pair3 p0 = split_pair_t<HP>(v.x, v.y, sa), p1 = split_pair_t<HP>(v.z, v.w, sa);
      char *d = buf + wa[i];
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kPlaneA) = i32x2{p0.h2, p1.h2};
      if (kNP == 3) *reinterpret_cast<i32x2 *>(d + 2 * kPlaneA) = i32x2{p0.h3, p1.h3};
    }
#pragma unroll
    for (int i = 0; i < NLB; ++i) {
      const float4 v = xb[set][i];
      const Pair3 p0 = split_pair_t<HP>(v.x, v.y, sb), p1 = split_pair_t<HP>(v.z, v.w, sb);
      char *d = buf + wb[i];
      *reinterpret_cast<i32x2 *>(d) = i32x2{p0.h1, p1.h1};
      *reinterpret_cast<i32x2 *>(d + kPlaneB) = i32x2{p0.h2, p1.h2};
      if (kNP == 3) *reinterpret_cast<i32x2 *>(d + 2 * kPlaneB) = i32x2{p0.h3, p1.h3};
    }
    if (s + NSET < ns) load_stage(s + NSET, set_c);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    i32x4 bf[NJW][3];
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      const int jt = NJW * wj + j;
      const char *r0 = buf + ((rdB + ((jt >> 2) << 8)) ^ ((jt & 3) << 6));
#pragma unroll
      for (int p = 0; p < kNP; ++p) bf[j][p] = tr_frag(r0 + p * kPlaneB, r0 + p * kPlaneB + 4 * kRowB);
    }
    // two accumulator rows per round (their A fragments live together): consecutive MFMAs are 2 NJW accumulators apart
    static_assert(NIW % 2 == 0, "accumulator rows are taken in pairs");
#pragma unroll
    for (int i = 0; i < NIW; i += 2) {
      i32x4 af[2][3];
#ifndef AMPCONV_WG_NOFENCE      // (the next round's reads are not hoisted into this round's MFMAs either)
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int it = NIW * wi + i + ii;
        const char *r0 = buf + ((rdA + ((it >> 2) << 8)) ^ ((it & 3) << 6));
#pragma unroll
        for (int p = 0; p < kNP; ++p) af[ii][p] = tr_frag(r0 + p * kPlaneA, r0 + p * kPlaneA + 4 * kRowA);
      }
      // Every fragment of the round is in its registers before the round's first MFMA issues, and no fragment read is
      // scheduled in among the MFMAs.  Left to itself hipcc interleaves them and re-uses an operand register for the next
      // fragment right behind the MFMA that reads it (`v_mfma .. v[128:131] ..; ds_read_b64_tr_b16 v[128:129] ..;
      // s_waitcnt lgkmcnt(0); v_mfma .. v[128:131]`); builds with that schedule gave WRONG sums in 7-100 % of the launches
      // of the masked product (one 16-bit element of a fragment stale; flags, rows and the LDS image verified correct
      // at the barrier by an in-kernel dump: DESIGN.md 4a), with this fence 0 of 1 600.  -DAMPCONV_WG_NOFENCE rebuilds it.
#ifndef AMPCONV_WG_NOFENCE
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // (+ 16 idle cycles behind the wait)
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int q = 0; q < Pl<HP>::NQ; ++q)
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int j = 0; j < NJW; ++j)
            acc[i + ii][j] = mfma_p<HP>(af[ii][kPA[HP][q]], bf[j][kPB[HP][q]], acc[i + ii][j]);
#ifndef AMPCONV_WG_NOFENCE
      // ... and the next round's (or stage's) reads do not redefine af[] / bf[] right behind the MFMA that reads them: the
      // last MFMA of the round has fetched its operands by then (8 idle cycles; tools/scan_tr_hazard.py --gate, rule WAR)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_nop 7" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  for (int s = 0; s < ns; s += NSET) {
    stage(s, std::integral_constant<int, 0>{});
    if (NSET == 2 && s + 1 < ns) stage(s + 1, std::integral_constant<int, NSET - 1>{});
  }

  // partial tile of this slice
  float *part = a.part + (size_t)slice * ((size_t)a.Nap * a.Nbp + a.Nap);
#pragma unroll
  for (int i = 0; i < NIW; ++i)
#pragma unroll
    for (int j = 0; j < NJW; ++j) {
      float *o = part + (size_t)(ti * TI + (NIW * wi + i) * 32 + 4 * fh) * a.Nbp + tj * TJ + (NJW * wj + j) * 32 + fr;
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(size_t)((e & 3) + 8 * (e >> 2)) * a.Nbp] = acc[i][j][e];
    }
  if (tj == 0) {
    // column sums of the A tile: kRowsA row-threads per float4 column, added through LDS in a fixed order
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float4 *red = reinterpret_cast<float4 *>(smem);
    red[ra * kColsA4 + ca] = cs;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t < kColsA4) {
      float4 sum = red[t];
#pragma unroll
      for (int r = 1; r < kRowsA; ++r) {
        const float4 v = red[r * kColsA4 + t];
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      *reinterpret_cast<float4 *>(part + (size_t)a.Nap * a.Nbp + ti * TI + 4 * t) = sum;
    }
  }
}

// flag[m] = 1.0 if the node of row m has an in-edge, else 0.0 (one thread per NODE: no division)
__global__ __launch_bounds__(256) void row_mask_kernel(const int32_t *__restrict__ rowptr, int L, int64_t n_nodes,
                                                       int64_t M, float *__restrict__ flag) {
  const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (n >= n_nodes) return;
  const float f = rowptr[n + 1] != rowptr[n] ? 1.f : 0.f;
  const int64_t m0 = n * L;
  for (int l = 0; l < L && m0 + l < M; ++l) flag[m0 + l] = f;
}

// out[e] = sum over the slices of part[s][e]; e < n_dw goes to dW, the rest to colsum.  256 threads = 32 float4
// elements x 8 slice phases: phase g adds slices g, g + 8, ... in order (four loads in flight), the eight phase sums
// meet in LDS and are added in a fixed order -- bitwise reproducible, and S / 8 dependent load rounds instead of S
// (scaled mode: amax_a / amax_b non-null, the dW part leaves through the two exact inverse scales)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, int S, int Na, int Nb, int Nap,
                                                           int Nbp, float *__restrict__ dW, float *__restrict__ colsum,
                                                           const float *__restrict__ amax_a,
                                                           const float *__restrict__ amax_b) {
  __shared__ float4 red[8][32];
  const int el = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t n_dw = (int64_t)Nap * Nbp, n_all = n_dw + Nap;        // the slabs cover the padded shape
  const int64_t e = ((int64_t)blockIdx.x * 32 + el) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < n_all) {
    float4 a1 = acc;
    int s = g;
    for (; s + 8 < S; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4 *>(part + (size_t)s * n_all + e);
      const float4 v1 = *reinterpret_cast<const float4 *>(part + (size_t)(s + 8) * n_all + e);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    }
    if (s < S) {
      const float4 v0 = *reinterpret_cast<const float4 *>(part + (size_t)s * n_all + e);
      acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
    }
    acc.x += a1.x; acc.y += a1.y; acc.z += a1.z; acc.w += a1.w;
  }
  red[g][el] = acc;
  __syncthreads();
  if (g == 0 && e < n_all) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float4 v = red[k][el];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (e < n_dw) {
      const int i = (int)(e / Nbp), j = (int)(e - (int64_t)i * Nbp);
      if (amax_a) {
        const float ua = plane_unscale(*amax_a), ub = plane_unscale(*amax_b);
        acc.x = acc.x * ua * ub; acc.y = acc.y * ua * ub; acc.z = acc.z * ua * ub; acc.w = acc.w * ua * ub;
      }
      if (i < Na && j < Nb) *reinterpret_cast<float4 *>(dW + (size_t)i * Nb + j) = acc;
    } else if (colsum && e - n_dw < Na) {
      *reinterpret_cast<float4 *>(colsum + (e - n_dw)) = acc;
    }
  }
}

struct WgradPlan {
  int S;
  int64_t rows_per_slice;
  int ti, tj;
};
inline WgradPlan wgrad_plan(int64_t M, int Nap, int Nbp) {       // padded shape (multiples of 128)
  WgradPlan p;
  static const bool small_ti = [] {                        // developer switch: 128 x 256 tiles of four waves, two per CU
    const char *e = getenv("AMPCONV_PROJ_WGRAD_TI");
    return e && atoi(e) == 128;
  }();
  p.tj = Nbp % 256 == 0 ? 256 : 128;
  p.ti = (Nap % 256 == 0 && p.tj == 256 && !small_ti) ? 256 : 128;       // 256 x 256: eight waves, one workgroup per CU
  const int64_t ntiles = (int64_t)(Nap / p.ti) * (Nbp / p.tj);
  const int64_t nstages = (M + kRS - 1) / kRS;
  // one round of workgroups: slices are dealt to the 8 XCDs in turn (round-robin dispatch) and every slice brings
  // `ntiles` workgroups, so an XCD's 32 CUs (x 2 for the 4-wave shapes) hold floor(32 / ntiles) slices each --
  // one slice more on some XCDs would run as a second round and double the time
  const int per_xcd = (p.ti == 256 ? 1 : 2) * 32;
  int64_t S = (int64_t)kXcd * (per_xcd / ntiles > 0 ? per_xcd / ntiles : 1);
  if (S < 1) S = 1;
  if (S > nstages) S = nstages > 0 ? nstages : 1;
  p.rows_per_slice = ((nstages + S - 1) / S) * kRS;
  p.S = (int)((M + p.rows_per_slice - 1) / p.rows_per_slice);
  if (p.S < 1) p.S = 1;
  return p;
}

}  // namespace

extern "C" size_t ampconv_proj_weight_image_bytes(int N, int K, int dtype) {
  if (dtype == AMPCONV_BF16) return ampconv_proj_weight_image_bytes_bf16(N, K);
  if (dtype != AMPCONV_F32 || N <= 0 || K <= 0) return 0;
  return image_amax_offset(N, K) + 512;      // padded shape: three bf16 planes, two fp16 planes, the weight's maximum
}

static bool supported_f32(int N, int K) { return N > 0 && K > 0 && N % 4 == 0 && K % 4 == 0; }

// fp32: rows and row strides are read as float4: every dimension a multiple of 4 (16-byte aligned rows); the tiles are
// padded internally (N to 128, K to 32), so embed_dim = 100 -- the reference's AMPGCN default -- is served too.
// bf16: multiples of 8 (the same 16 bytes)
extern "C" int ampconv_proj_supported(int N, int K, int dtype) {
  if (dtype == AMPCONV_BF16) return ampconv_proj_supported_bf16(N, K) ? 1 : 0;
  return dtype == AMPCONV_F32 && supported_f32(N, K);
}

extern "C" int ampconv_proj_weight_images(int count, const ampconv_weight_image_t *jobs, int dtype, void *stream) {
  if (count < 0 || count > 8 || (count && !jobs)) return AMPCONV_E_BADARG;
  if (count == 0) return AMPCONV_OK;
  if (dtype == AMPCONV_BF16) return ampconv_proj_weight_images_bf16(count, jobs, (hipStream_t)stream);
  if (dtype != AMPCONV_F32) return AMPCONV_E_DTYPE;
  ImageJobs js;
  int most = 0;
  for (int i = 0; i < count; ++i) {
    const ampconv_weight_image_t &w = jobs[i];
    if (!supported_f32(w.N, w.K) || !w.W || !w.image || (uintptr_t)w.image % 16) return AMPCONV_E_BADARG;
    js.j[i] = ImageJob{(const float *)w.W, w.stride_n, w.stride_k, w.N, w.K, (char *)w.image};
    const int total = ((w.N + 127) / 128 * 128) * (((w.K + 31) / 32 * 32) / 8);
    most = total > most ? total : most;
  }
  weight_absmax_kernel<<<dim3(kAmaxParts, count), 256, 0, (hipStream_t)stream>>>(js);
  weight_image_kernel<<<dim3((most + 255) / 256, count), 256, 0, (hipStream_t)stream>>>(js);
  return ampconv_launch_status();
}

extern "C" int ampconv_proj_weight_image(const void *W, int64_t stride_n, int64_t stride_k, int N, int K,
                                         void *image, int dtype, void *stream) {
  const ampconv_weight_image_t job{W, stride_n, stride_k, N, K, image};
  return ampconv_proj_weight_images(1, &job, dtype, stream);
}

// fp32 storage; out_bound != null: plane output (ampconv_proj_rows_planes)
static int proj_rows_f32(const void *A_, int64_t lda, int64_t M, int K, const void *wimage, int N, const void *bias_,
                         const int32_t *rowptr, int L, void *out_, int64_t ldc, const float *a_absmax, float *out_absmax,
                         int amax_col0, const float *out_bound, int plane_dh, int row_scale, void *stream) {
  if (out_absmax && !a_absmax) return AMPCONV_E_BADARG;           // recorded by the scaled kernels only
  // planes: scaled mode, whole 128-byte slots, and the unpadded tile shapes only (N % 128 == 0, K % 32 == 0)
  if (out_bound && (!a_absmax || N % 128 || K % 32 || ldc % 32)) return AMPCONV_E_BADARG;
  if (row_scale && !rowptr) return AMPCONV_E_BADARG;
  const float *A = (const float *)A_, *bias = (const float *)bias_;
  float *out = (float *)out_;
  if (M < 0 || !supported_f32(N, K) || lda < K || ldc < N || lda % 4) return AMPCONV_E_BADARG;
  if (M == 0) return AMPCONV_OK;
  if (!A || !wimage || !out || (uintptr_t)A % 16 || (uintptr_t)wimage % 16) return AMPCONV_E_BADARG;
  if (rowptr && (L <= 0 || M > 0x7fffffff)) return AMPCONV_E_BADARG;
  if (ldc % 4 || (uintptr_t)out % 16) return AMPCONV_E_BADARG;
  // tile shape (developer switch AMPCONV_PROJ_ROWS: 0 = 128 x 256 tile of 4 waves where N allows, 1 = 256 x 256 of 8)
  static const int variant = [] {
    const char *e = getenv("AMPCONV_PROJ_ROWS");
    return e ? atoi(e) : 0;
  }();
  const int Np = (N + 127) / 128 * 128, Kp = (K + 31) / 32 * 32;
  const bool ragged = Np != N || Kp != K;
  const int shape = (Np % 256 || variant == 2) ? 2 : variant;      // 0: 128 x 256 / 4 waves, 1: 256 x 256 / 8 waves, 2: 128 x 128 / 4 waves
  const int bm = shape == 1 ? 256 : 128, bn = shape == 2 ? 128 : 256;
  const int64_t rts = (M + bm - 1) / bm;
  if (rts > (int64_t)INT32_MAX / 64) return AMPCONV_E_BADARG;
  const int64_t rtp = (rts + 7) / 8 * 8;
  const bool hp = a_absmax != nullptr;                 // scaled two-plane mode: the image's fp16 half and its maximum
  const char *img = (const char *)wimage;
  RowsArgs a{A, lda, M, K, N, hp ? img + image_half_offset(N, K) : img, bias, rowptr, L, out, ldc, (int)rts,
             rtp * (Np / bn), Kp, Np, a_absmax, (const float *)(img + image_amax_offset(N, K)) + kAmaxParts, out_absmax,
             amax_col0, out_bound, plane_dh, row_scale};
  const int n_cu = cu_count();
  // a multiple of 8: slot u of a workgroup keeps u % 8 (its XCD label), so "my next slot is invalid" means "nothing
  // further for me" only then (a.tiles is a multiple of 8 by construction)
  int64_t grid = (int64_t)n_cu * (shape == 1 ? 1 : (shape == 2 && hp ? AMPCONV_PROJ_ROWS_128_WAVES : 2)) / kXcd * kXcd;
  if (grid < kXcd) grid = kXcd;
  if (grid > a.tiles) grid = a.tiles;
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)grid;
#define ROWS_LAUNCH(BM_, BN_, WM_, WN_, NT_)                                                          \
  do {                                                                                                \
    if (out_bound) proj_rows_kernel<BM_, BN_, WM_, WN_, false, true, true><<<g, NT_, 0, st>>>(a); \
    else if (hp && ragged) proj_rows_kernel<BM_, BN_, WM_, WN_, true, true><<<g, NT_, 0, st>>>(a);     \
    else if (hp) proj_rows_kernel<BM_, BN_, WM_, WN_, false, true><<<g, NT_, 0, st>>>(a);              \
    else if (ragged) proj_rows_kernel<BM_, BN_, WM_, WN_, true, false><<<g, NT_, 0, st>>>(a);          \
    else proj_rows_kernel<BM_, BN_, WM_, WN_, false, false><<<g, NT_, 0, st>>>(a);                     \
  } while (0)
  if (shape == 0) ROWS_LAUNCH(128, 256, 2, 2, 256);
  else if (shape == 1) ROWS_LAUNCH(256, 256, 2, 4, 512);
  else ROWS_LAUNCH(128, 128, 2, 2, 256);
#undef ROWS_LAUNCH
  return ampconv_launch_status();
}

extern "C" int ampconv_proj_rows(const void *A_, int64_t lda, int64_t M, int K, const void *wimage, int N,
                                 const void *bias_, const int32_t *rowptr, int L, void *out_, int64_t ldc,
                                 const int32_t *nodes, int64_t n_nodes, const float *a_absmax, float *out_absmax,
                                 int dtype, void *stream) {
  if (dtype == AMPCONV_BF16) {
    if (a_absmax || out_absmax) return AMPCONV_E_DTYPE;           // scaled mode: fp32 storage only
    return ampconv_proj_rows_bf16(A_, lda, M, K, wimage, N, bias_, rowptr, L, out_, ldc, nodes, n_nodes,
                                  (hipStream_t)stream);
  }
  if (dtype != AMPCONV_F32 || nodes) return AMPCONV_E_DTYPE;      // node lists: bf16 storage only
  return proj_rows_f32(A_, lda, M, K, wimage, N, bias_, rowptr, L, out_, ldc, a_absmax, out_absmax, 0, nullptr, 32, 0, stream);
}

extern "C" int ampconv_proj_rows_planes(const void *A, int64_t lda, int64_t M, int K, const void *wimage, int N,
                                        const void *bias, const int32_t *rowptr, int L, int row_scale, void *out,
                                        int64_t ldc, const float *a_absmax, const float *out_bound, float *out_absmax,
                                        int absmax_col0, int plane_dh, void *stream) {
  if (!out_bound || !a_absmax || row_scale < 0 || row_scale > 1 || absmax_col0 < 0 || (plane_dh != 32 && plane_dh != 16))
    return AMPCONV_E_BADARG;
  return proj_rows_f32(A, lda, M, K, wimage, N, bias, rowptr, L, out, ldc, a_absmax, out_absmax, absmax_col0, out_bound,
                       plane_dh, row_scale, stream);
}

// bound of the magnitudes of out = A W^T + bias from the largest magnitude of A: a_absmax * max_n sum_k |W[n][k]| +
// max_n |bias[n]| (W[n][k] at W[n * stride_n + k * stride_k]) -- what the plane output of proj_rows scales by, known
// BEFORE the product runs.  One workgroup: the weights are a few hundred KB.
namespace {
__global__ __launch_bounds__(1024) void out_bound_kernel(const float *__restrict__ W, int64_t sn, int64_t sk, int N, int K,
                                                         const float *__restrict__ bias, const float *__restrict__ amax,
                                                         float *__restrict__ out) {
  __shared__ float red[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float l1 = 0.f, bm = 0.f;
  for (int n = w; n < N; n += 16) {
    float sum = 0.f;
    for (int k = lane; k < K; k += 64) sum += finite_abs(W[(int64_t)n * sn + (int64_t)k * sk]);
    l1 = fmaxf(l1, wave_sum(sum));
  }
  if (bias)
    for (int n = threadIdx.x; n < N; n += 1024) bm = fmaxf(bm, finite_abs(bias[n]));
  bm = wave_max(bm);
  if (lane == 0) { red[0][w] = l1; red[1][w] = bm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) { l1 = fmaxf(l1, red[0][i]); bm = fmaxf(bm, red[1][i]); }
    out[0] = *amax * l1 + bm;
  }
}
}  // namespace

extern "C" int ampconv_proj_out_bound(const void *W, int64_t stride_n, int64_t stride_k, int N, int K, const void *bias,
                                      const float *a_absmax, float *out, void *stream) {
  if (!W || !a_absmax || !out || N <= 0 || K <= 0) return AMPCONV_E_BADARG;
  out_bound_kernel<<<1, 1024, 0, (hipStream_t)stream>>>((const float *)W, stride_n, stride_k, N, K, (const float *)bias,
                                                        a_absmax, out);
  return ampconv_launch_status();
}

// plane slots -> fp32, in place layout (the reverse of the PLANES epilogue; the lazily served side outputs and the
// fall-back paths read the projection buffer as fp32): X[M, K] with K a multiple of 32, rows ld floats apart
namespace {
__global__ __launch_bounds__(256) void planes_to_f32_kernel(const char *__restrict__ X, int64_t ld, int64_t M, int K8, int dh,
                                                            const float *__restrict__ bound, float *__restrict__ out,
                                                            int64_t ldo) {
  const float u = 1.f / plane_scale(*bound);
  const int pps = dh / 8;                                  // 8-channel pieces per slot
  const int64_t total = M * K8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / K8;
    const int r = (int)(i - row * K8), slot = r / pps, c = r - slot * pps;
    const char *src = X + (row * ld + (int64_t)slot * dh) * 4 + 16 * c;
    const uint4 h = *reinterpret_cast<const uint4 *>(src), l = *reinterpret_cast<const uint4 *>(src + 2 * dh);
    const unsigned hu[4] = {h.x, h.y, h.z, h.w}, lu[4] = {l.x, l.y, l.z, l.w};
    float v[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f16x2 hv = __builtin_bit_cast(f16x2, hu[k]), lv = __builtin_bit_cast(f16x2, lu[k]);
      v[2 * k] = ((float)hv[0] + (float)lv[0]) * u;
      v[2 * k + 1] = ((float)hv[1] + (float)lv[1]) * u;
    }
    float *o = out + row * ldo + slot * dh + 8 * c;
    *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}
}  // namespace

extern "C" int ampconv_planes_to_f32(const void *X, int64_t ld, int64_t M, int K, int plane_dh, const float *bound, void *out,
                                     int64_t ldo, void *stream) {
  if (M < 0 || K <= 0 || (plane_dh != 32 && plane_dh != 16) || K % plane_dh || ld < K || ldo < K || ld % 4 || ldo % 4 || !bound)
    return AMPCONV_E_BADARG;
  if (M == 0) return AMPCONV_OK;
  if (!X || !out || (uintptr_t)X % 16 || (uintptr_t)out % 16) return AMPCONV_E_BADARG;
  const int64_t pieces = M * (K / 8);
  int64_t grid = (pieces + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 16;
  if (grid > cap) grid = cap;
  planes_to_f32_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>((const char *)X, ld, M, K / 8, plane_dh, bound,
                                                                        (float *)out, ldo);
  return ampconv_launch_status();
}

// largest finite magnitude of X[M, K] (row stride ld), merged into *out by an atomic max: zero it first (reset != 0
// does) or let several calls accumulate.  NaN and infinities are skipped: they propagate through the products by
// themselves, in the rows they sit in.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T *__restrict__ X, int64_t ld, int64_t M, int K4,
                                                     float *__restrict__ out) {
  // K4 = 16-byte pieces per row; a workgroup walks pieces blockIdx.x * 256 + t, + gridDim.x * 256, ...
  constexpr int EPP = 16 / sizeof(T);
  const int64_t total = M * K4;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / K4;
    const int c = (int)(i - row * K4);
    const uint4 raw = *reinterpret_cast<const uint4 *>(X + row * ld + (int64_t)c * EPP);
    if constexpr (sizeof(T) == 4) {
      m = fmaxf(m, fmaxf(fmaxf(finite_abs(__builtin_bit_cast(float, raw.x)), finite_abs(__builtin_bit_cast(float, raw.y))),
                         fmaxf(finite_abs(__builtin_bit_cast(float, raw.z)), finite_abs(__builtin_bit_cast(float, raw.w)))));
    } else {
      const unsigned u[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        m = fmaxf(m, fmaxf(finite_abs(lo_as_f32(u[k])), finite_abs(hi_as_f32(u[k]))));
    }
  }
  __shared__ float red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(reinterpret_cast<unsigned *>(out), __builtin_bit_cast(unsigned, m));
  }
}
}  // namespace

extern "C" int ampconv_absmax(const void *X, int64_t ld, int64_t M, int K, int dtype, float *out, int reset,
                              void *stream) {
  if (dtype != AMPCONV_F32 && dtype != AMPCONV_BF16) return AMPCONV_E_DTYPE;
  const int epp = dtype == AMPCONV_F32 ? 4 : 8;
  if (!out || M < 0 || K < 0 || K % epp || ld < K || ld % epp) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (reset) {
    const hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
  }
  if (M == 0 || K == 0) return AMPCONV_OK;
  if (!X || (uintptr_t)X % 16) return AMPCONV_E_BADARG;
  const int64_t pieces = M * (K / epp);
  int64_t grid = (pieces + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 16;         // eight loads per thread and more: the atomics stay few
  if (grid > cap) grid = cap;
  if (dtype == AMPCONV_F32)
    absmax_kernel<float><<<(unsigned)grid, 256, 0, st>>>((const float *)X, ld, M, K / epp, out);
  else
    absmax_kernel<unsigned short><<<(unsigned)grid, 256, 0, st>>>((const unsigned short *)X, ld, M, K / epp, out);
  return ampconv_launch_status();
}

// the same pass with a RANGE statistic beside the maximum (fp32): out[0] = largest finite magnitude, out[1] = the smallest
// non-zero maximum of any group of 8 consecutive 16-byte pieces (32 channels: one head slot of a row where K % 32 == 0).
// out[1] far below out[0] means whole rows / heads live binades under the tensor's maximum -- the data on which ONE scale
// per tensor costs the small rows their low plane (include/ampconv.h "SCALED MODE"); the caller then takes the exact
// kernels.  Single elements near zero do not count (absolute error is what a sum feels), all-zero groups neither.
namespace {
__global__ void absmax_stats_init_kernel(float *out) {
  out[0] = 0.f;
  out[1] = __builtin_inff();
}
template <int CTRL>
__device__ __forceinline__ float dpp_get(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__global__ __launch_bounds__(256) void absmax_stats_kernel(const float *__restrict__ X, int64_t ld, int64_t M, int K4,
                                                           float *__restrict__ out) {
  const int64_t total = M * K4;
  float m = 0.f, smin = __builtin_inff();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / K4;
    const int c = (int)(i - row * K4);
    const float4 v = *reinterpret_cast<const float4 *>(X + row * ld + (int64_t)c * 4);
    const float m4 = finite_abs_max(0.f, v);
    m = fmaxf(m, m4);
    float g = fmaxf(m4, dpp_get<0xB1>(m4));     // quad_perm [1,0,3,2]
    g = fmaxf(g, dpp_get<0x4E>(g));             // quad_perm [2,3,0,1]
    g = fmaxf(g, dpp_get<0x141>(g));            // row_half_mirror: the 8 lanes of a half row (masked lanes read as 0)
    if (g > 0.f) smin = fminf(smin, g);
  }
  __shared__ float red[2][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    m = fmaxf(m, __shfl_xor(m, o));
    smin = fminf(smin, __shfl_xor(smin, o));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = m;
    red[1][threadIdx.x >> 6] = smin;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    smin = fminf(fminf(red[1][0], red[1][1]), fminf(red[1][2], red[1][3]));
    atomicMax(reinterpret_cast<unsigned *>(out), __builtin_bit_cast(unsigned, m));          // non-negative floats order
    atomicMin(reinterpret_cast<unsigned *>(out + 1), __builtin_bit_cast(unsigned, smin));   // as their bits
  }
}
}  // namespace

extern "C" int ampconv_absmax_stats(const void *X, int64_t ld, int64_t M, int K, float *out, void *stream) {
  if (!out || M < 0 || K < 0 || K % 4 || ld < K || ld % 4) return AMPCONV_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  absmax_stats_init_kernel<<<1, 1, 0, st>>>(out);
  if (M == 0 || K == 0) return ampconv_launch_status();
  if (!X || (uintptr_t)X % 16) return AMPCONV_E_BADARG;
  const int64_t pieces = M * (K / 4);
  int64_t grid = (pieces + 255) / 256;
  const int64_t cap = (int64_t)cu_count() * 16;
  if (grid > cap) grid = cap;
  absmax_stats_kernel<<<(unsigned)grid, 256, 0, st>>>((const float *)X, ld, M, K / 4, out);
  return ampconv_launch_status();
}

extern "C" size_t ampconv_proj_wgrad_workspace_bytes(int64_t M, int Na, int Nb, int dtype) {
  if (dtype == AMPCONV_BF16) return ampconv_proj_wgrad_workspace_bytes_bf16(M, Na, Nb);
  if (dtype != AMPCONV_F32 || M < 0 || Na <= 0 || Nb <= 0 || Na % 4 || Nb % 4) return 0;
  const int Nap = (Na + 127) / 128 * 128, Nbp = (Nb + 127) / 128 * 128;
  const WgradPlan p = wgrad_plan(M, Nap, Nbp);
  // [S partial slabs][one flag per row: used with a mask]
  return (size_t)p.S * ((size_t)Nap * Nbp + Nap) * sizeof(float) + ((size_t)M + 3) / 4 * 16;
}

extern "C" int ampconv_proj_wgrad(const void *A_, int64_t lda, const void *B_, int64_t ldb, int64_t M, int Na,
                                  int Nb, const int32_t *rowptr, int L, void *dW_, void *colsum_,
                                  void *workspace, size_t workspace_bytes, const int32_t *nodes, int64_t n_nodes,
                                  const float *a_absmax, const float *b_absmax, int dtype, void *stream) {
  if (dtype == AMPCONV_BF16) {
    if (a_absmax || b_absmax) return AMPCONV_E_DTYPE;             // scaled mode: fp32 storage only
    return ampconv_proj_wgrad_bf16(A_, lda, B_, ldb, M, Na, Nb, rowptr, L, dW_, colsum_, workspace, workspace_bytes,
                                   nodes, n_nodes, (hipStream_t)stream);
  }
  if ((a_absmax != nullptr) != (b_absmax != nullptr)) return AMPCONV_E_BADARG;      // both operands or neither
  if (dtype != AMPCONV_F32 || nodes) return AMPCONV_E_DTYPE;      // node lists: bf16 storage only
  const float *A = (const float *)A_, *B = (const float *)B_;
  float *dW = (float *)dW_, *colsum = (float *)colsum_;
  if (M < 0 || Na <= 0 || Nb <= 0 || Na % 4 || Nb % 4 || lda < Na || ldb < Nb || lda % 4 || ldb % 4)
    return AMPCONV_E_BADARG;
  if (!dW || (uintptr_t)dW % 16 || (colsum && (uintptr_t)colsum % 16)) return AMPCONV_E_BADARG;
  if (rowptr && (L <= 0 || M > 0x7fffffff)) return AMPCONV_E_BADARG;       // (rows of a masked product are addressed as int32)
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) {
    hipError_t e = hipMemsetAsync(dW, 0, sizeof(float) * (size_t)Na * Nb, st);
    if (e == hipSuccess && colsum) e = hipMemsetAsync(colsum, 0, sizeof(float) * Na, st);
    return e == hipSuccess ? AMPCONV_OK : (int)e;
  }
  if (!A || !B || (uintptr_t)A % 16 || (uintptr_t)B % 16 || !workspace || (uintptr_t)workspace % 16)
    return AMPCONV_E_BADARG;
  const int Nap = (Na + 127) / 128 * 128, Nbp = (Nb + 127) / 128 * 128;
  const bool ragged = Nap != Na || Nbp != Nb;
  const WgradPlan p = wgrad_plan(M, Nap, Nbp);
  const size_t n_all = (size_t)Nap * Nbp + Nap;
  const size_t slab_bytes = (size_t)p.S * n_all * sizeof(float);
  if (workspace_bytes < slab_bytes + (rowptr ? ((size_t)M + 3) / 4 * 16 : 0)) return AMPCONV_E_WORKSPACE;
  float *rowflag = rowptr ? (float *)((char *)workspace + slab_bytes) : nullptr;
  if (rowptr) {
    const int64_t n_nodes = (M + L - 1) / L;
    row_mask_kernel<<<(unsigned)((n_nodes + 255) / 256), 256, 0, st>>>(rowptr, L, n_nodes, M, rowflag);
  }
  WgradArgs a{A, lda, B, ldb, M, Na, Nb, rowflag, (float *)workspace, p.S, p.rows_per_slice, Nap, Nbp, a_absmax,
              b_absmax};
  const bool hp = a_absmax != nullptr;
  const int ntiles = (Nap / p.ti) * (Nbp / p.tj);
  const unsigned grid = (unsigned)(((p.S + 7) / 8 * 8) * ntiles);
#define WGRAD_LAUNCH(TI_, TJ_, WI_, WJ_, NT_)                                                              \
  do {                                                                                                     \
    if (hp) {                                                                                              \
      if (rowptr && ragged) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, true, true, true><<<grid, NT_, 0, st>>>(a);      \
      else if (rowptr) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, true, false, true><<<grid, NT_, 0, st>>>(a);          \
      else if (ragged) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, false, true, true><<<grid, NT_, 0, st>>>(a);          \
      else proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, false, false, true><<<grid, NT_, 0, st>>>(a);                     \
    } else if (rowptr && ragged) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, true, true, false><<<grid, NT_, 0, st>>>(a); \
    else if (rowptr) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, true, false, false><<<grid, NT_, 0, st>>>(a);           \
    else if (ragged) proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, false, true, false><<<grid, NT_, 0, st>>>(a);           \
    else proj_wgrad_kernel<TI_, TJ_, WI_, WJ_, false, false, false><<<grid, NT_, 0, st>>>(a);                      \
  } while (0)
  if (p.ti == 256) WGRAD_LAUNCH(256, 256, 2, 4, 512);
  else if (p.tj == 256) WGRAD_LAUNCH(128, 256, 2, 2, 256);
  else WGRAD_LAUNCH(128, 128, 2, 2, 256);
#undef WGRAD_LAUNCH
  wgrad_reduce_kernel<<<(unsigned)((n_all / 4 + 31) / 32), 256, 0, st>>>((const float *)workspace, p.S, Na, Nb, Nap, Nbp,
                                                                         dW, colsum, a_absmax, b_absmax);
  return ampconv_launch_status();
}

#ifdef AMPCONV_PROJ_STAMPS
// diagnostic build only (tools/stamp_proj.py)
extern "C" int ampconv_debug_read_proj_stamps(unsigned long long *out, int n) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_proj_stamps), sizeof(unsigned long long) * (size_t)n);
}
#endif
